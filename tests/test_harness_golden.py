"""Host-side mirror of the reference harness vs golden strings captured by importing the
reference's process_vdif.py (tests/golden/make_harness_golden.py)."""
import contextlib
import io
import json
import os
import stat

import pytest

from frb_baseband_amd import process_vdif as pv

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "harness_golden.json")))


@pytest.mark.parametrize("case", GOLD["make_hdr"], ids=lambda c: c["args"]["base"])
def test_make_hdr_text(case, tmp_path):
    a = case["args"]
    fn = os.path.join(str(tmp_path), a["base"])
    path = pv.make_hdr(a["psr"], a["freq"], fn, pol=a["pol"], usb=a["usb"], ra=a["ra"], dec=a["dec"], bw=a["bw"],
                       telescope=a["telescope"], tmp=a["tmp"])
    try:
        text = open(path).read()
        assert text.replace(str(tmp_path), "<D>") == case["text"]
        assert not text.endswith("\n")
        if a["tmp"]:
            assert path == case["hdr_relpath"]
        else:
            assert os.path.relpath(path, str(tmp_path)) == case["hdr_relpath"]
    finally:
        if a["tmp"]:
            os.remove(path)


@pytest.mark.parametrize("case", GOLD["run_digifil"], ids=lambda c: c["cmd"][-40:])
def test_digifil_command_string(case):
    kw = dict(case["kwargs"])
    out = case["returns"]
    cmd = pv.digifil_command(case["hdr"], out, kw.get("start", 1), kw.get("nsecs", 120), kw.get("nchan", 128),
                             kw.get("pol", 2), kw.get("nbit", 8), kw.get("tscrunch", 1), kw.get("nthreads", 1),
                             kw.get("dm", 0.0), kw.get("coherent", False), kw.get("keepBP", False))
    assert cmd == case["cmd"]


@pytest.mark.parametrize("case", GOLD["errors"], ids=lambda c: str(c["kwargs"]))
def test_input_errors(case):
    with pytest.raises(pv.InputError) as ei, contextlib.redirect_stdout(io.StringIO()):
        pv.run_digifil("/d/x.hdr", "/fifo", overwrite=True, **case["kwargs"])
    assert ei.value.message == case["message"]


@pytest.mark.parametrize("case", GOLD["argparse"], ids=lambda c: c["argv"][:40])
def test_argparse_namespace(case):
    ns = pv.options(case["argv"].split()[1:])
    assert vars(ns) == case["namespace"]


def test_existing_output_rules(tmp_path):
    hdr = str(tmp_path / "x.vdif_pol2.hdr")
    fil = str(tmp_path / "x.vdif_pol2.fil")
    open(fil, "w").write("old")
    with pytest.raises(pv.InputError):                     # process_vdif.py:150-152
        pv.run_digifil(hdr, None, overwrite=False)
    os.remove(fil)
    os.mkfifo(fil)                                         # a FIFO must survive --force (:146-149)
    with pytest.raises(pv.RunError), contextlib.redirect_stdout(io.StringIO()):
        pv.run_digifil(hdr, None, overwrite=True)          # dies later (no hdr / no GPU), FIFO kept
    assert stat.S_ISFIFO(os.stat(fil).st_mode)


def test_main_requires_sideband():
    with pytest.raises(pv.InputError):
        pv.main(["B0329+54", "x.vdif"])
    with pytest.raises(pv.InputError):
        pv.main(["B0329+54", "x.vdif", "-u", "-l"])
