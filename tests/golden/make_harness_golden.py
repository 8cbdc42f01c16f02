#!/usr/bin/env python3
"""Generate tests/golden/harness_golden.json by importing the REFERENCE harness
(/root/reference/process_vdif.py) in the build container and recording what it produces:
the .hdr text of make_hdr, the digifil command lines of run_digifil (subprocess.check_call is
stubbed because digifil itself is not installed), the argparse namespaces and the error messages.

Run here only (the reference does not travel to the GPU box); the JSON fixture is data
(inputs and expected outputs), not reference source.
    python3 -B tests/golden/make_harness_golden.py
"""
import importlib.util
import json
import os
import sys
import tempfile

REF = "/root/reference/process_vdif.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_golden.json")


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_process_vdif", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref()
    golden = {"make_hdr": [], "run_digifil": [], "errors": [], "argparse": []}

    # ---- make_hdr -----------------------------------------------------------------------------
    hdr_cases = [
        dict(psr="R3", freq=1340.49, base="pr001a_ef_no0001_IF1.vdif", pol=2, usb=False,
             ra="01:58:00.7502", dec="65:43:00.3152", bw=32.0, telescope="effelsberg", tmp=False),
        dict(psr="B0329+54", freq=1608.0, base="x_IF2.vdif", pol=4, usb=True,
             ra="03:32:59.4", dec="54:34:43.6", bw=16.0, telescope="ONSALA85", tmp=False),
        dict(psr="R3", freq=1404.49, base="y_IF3.vdif", pol=0, usb=False,
             ra="01:58:00.7502", dec="65:43:00.3152", bw=64.0, telescope="srt", tmp=True),
    ]
    with tempfile.TemporaryDirectory() as d:
        for c in hdr_cases:
            fn = os.path.join(d, c["base"])
            path = ref.make_hdr(c["psr"], c["freq"], fn, pol=c["pol"], usb=c["usb"], ra=c["ra"], dec=c["dec"],
                                bw=c["bw"], telescope=c["telescope"], tmp=c["tmp"])
            with open(path) as f:
                text = f.read()
            golden["make_hdr"].append({
                "args": c,
                "hdr_relpath": path if c["tmp"] else os.path.relpath(path, d),
                "text": text.replace(d, "<D>"),
            })
            if c["tmp"]:
                os.remove(path)

    # ---- run_digifil ----------------------------------------------------------------------------
    recorded = []

    def fake_check_call(cmd, shell=True, stdout=None, stderr=None):
        recorded.append(cmd)
        return 0

    ref.subprocess.check_call = fake_check_call
    cases = [
        dict(start=0, nsecs=10, nchan=128, pol=2, nbit=8),
        dict(start=0, nsecs=10, nchan=1024, pol=2, nbit=8),
        dict(start=0, nsecs=10, nchan=1024, pol=4, nbit=8),
        dict(start=0, nsecs=10, nchan=4096, pol=2, nbit=8, tscrunch=8),
        dict(start=0, nsecs=10, nchan=2048, pol=2, nbit=8, dm=56.7, coherent=True),
        dict(start=1, nsecs=120, nchan=512, pol=0, nbit=8),
        dict(start=1, nsecs=120, nchan=512, pol=1, nbit=2),
        dict(start=1, nsecs=120, nchan=64, pol=3, nbit=-32, keepBP=True),
        dict(start=1, nsecs=120, nchan=256, pol=2, nbit=16, dm=26.8),
        dict(start=0.0, nsecs=10.0, nchan=1024, pol=2, nbit=8, nthreads=1, tscrunch=1),
    ]
    import contextlib
    import io
    for kw in cases:
        recorded.clear()
        hdr = "/d/x_IF1.vdif_pol%d.hdr" % kw["pol"]
        with contextlib.redirect_stdout(io.StringIO()):
            ret = ref.run_digifil(hdr, "/fifo", overwrite=True, **kw)
        golden["run_digifil"].append({"hdr": hdr, "fil_out_dir": "/fifo", "kwargs": kw, "cmd": recorded[0],
                                      "returns": ret})
    # as the real CLI calls it (note the '//')
    recorded.clear()
    hdr = "/scratch0/u/pr001a/pr001a_ef_no0001_IF1.vdif_pol2.hdr"
    with contextlib.redirect_stdout(io.StringIO()):
        ret = ref.run_digifil(hdr, "/tmp/u/fifos/", 0.0, 10.0, 1024, overwrite=True, pol=2, nbit=8, tscrunch=1,
                              nthreads=1, keepBP=False)
    golden["run_digifil"].append({"hdr": hdr, "fil_out_dir": "/tmp/u/fifos/",
                                  "kwargs": dict(start=0.0, nsecs=10.0, nchan=1024, pol=2, nbit=8, tscrunch=1,
                                                 nthreads=1, keepBP=False),
                                  "cmd": recorded[0], "returns": ret})

    # ---- errors ---------------------------------------------------------------------------------
    for kw in (dict(pol=5), dict(nbit=4)):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref.run_digifil("/d/x.hdr", "/fifo", overwrite=True, **kw)
            golden["errors"].append({"kwargs": kw, "error": None})
        except ref.InputError as exc:
            golden["errors"].append({"kwargs": kw, "error": "InputError", "message": exc.message})

    # ---- argparse surface -----------------------------------------------------------------------
    argvs = [
        "process_vdif R3 --ra 01:58:00.7502 --dec=65:43:00.3152 /scratch0/u/pr001a/pr001a_ef_no0001_IF1.vdif "
        "-f 1340.49 -b 32.0 -l --nchan 1024 --nsec 10 --start 0 --force -t effelsberg --pol 2 --nthreads 1 "
        "--tscrunch 1 --fil_out_dir /tmp/u/fifos/ --nbit=8",
        "process_vdif B0329+54 x.vdif -u",
        "process_vdif B0329+54 x.vdif -l --keepBP --pol 4 --nbit=-32 --tscrunch 8 --hdr_only",
    ]
    for a in argvs:
        old = sys.argv
        sys.argv = a.split()
        try:
            ns = ref.options()
        finally:
            sys.argv = old
        golden["argparse"].append({"argv": a, "namespace": vars(ns)})

    with open(OUT, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
