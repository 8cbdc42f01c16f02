#!/usr/bin/env python3
"""Per-IF channeliser harness: host-side mirror of the reference's process_vdif.py.

Same command line (process_vdif.py:9-99), same ``make_hdr`` side file (:115-139), same
``run_digifil`` signature, output naming, FIFO handling and error classes (:142-199, :232-253) --
but ``run_digifil`` drives the MI355X channeliser through the C ABI (include/frbch.h) instead of
launching DSPSR's digifil.  The digifil command string is still assembled (and printed) exactly as
the reference does, because it is the contract the C-side parser is tested against.

backend = "abi"  in-process ctypes call into csrc/libfrbch.so (default of ``run_digifil`` as a library call)
backend = "shim" runs csrc/digifil (our flag-compatible executable) as a subprocess, which keeps a
                 process whose name contains "digifil" alive per IF for base2fil.sh's `pwait`
                 throttle (base2fil.sh:21-28).  DEFAULT OF THE COMMAND LINE (``main``), because that is how
                 base2fil.sh launches this file (:61-64); FRBCH_BACKEND=abi selects the in-process call.

Extensions (not in the reference; the argparse surface stays the reference's, golden-tested): FRBCH_IQUV=1 with
``--pol 4`` writes Stokes I,Q,U,V instead of the coherency products (``run_digifil(..., iquv=True)``, shim flag
``-iquv``); FRBCH_DEVICE=<n> picks the GPU; with ``--do_prepdata`` FRBCH_PREP=gpu dedisperses on the GPU
(``prep.prepdata_gpu``) instead of launching PRESTO.
"""
from __future__ import annotations

import argparse
import os
import random
import stat
import string
import subprocess

from .channeliser import Channeliser, Error, InputError, RunError  # noqa: F401  (re-exported names)

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(_HERE, "csrc", "digifil")
VALID_NBIT = [2, 8, 16, -32]


def options(argv=None):
    """argparse surface of process_vdif.py:9-99 (same flags, defaults and destinations)."""
    parser = argparse.ArgumentParser()
    general = parser.add_argument_group("General info about the data.")
    digifil = parser.add_argument_group("Input to digifil.")
    prep = parser.add_argument_group("Input to prepdata/prepsubband")
    general.add_argument("psrname", type=str, help="B- or J-name of the target; unknown sources also need --ra/--dec.")
    general.add_argument("filename", type=str, help="name of the raw vdif file")
    general.add_argument("-f", "--freq", type=float, default=1608.0, help="CENTRAL frequency in MHz. Default=%(default)s MHz")
    general.add_argument("--ra", type=str, default=None, help="RA hh:mm:ss.ss (required if psrname is not a known pulsar)")
    general.add_argument("--dec", type=str, default=None, help="Dec dd:mm:ss.ss (required if psrname is not a known pulsar)")
    general.add_argument("-b", "--bw", type=float, default=16.0, help="Bandwidth of the scan. Default=%(default)s MHz")
    general.add_argument("-u", "--usb", action="store_true", help="upper side band (either -u or -l MUST be set)")
    general.add_argument("-l", "--lsb", action="store_true", help="lower side band (either -u or -l MUST be set)")
    general.add_argument("-t", "--telescope", type=str, default="ONSALA85", help="tempo/tempo2 telescope name. Default=%(default)s")
    general.add_argument("--use_tmp", action="store_true", help="put intermediate files in /tmp")
    general.add_argument("--hdr_only", action="store_true", help="only create the hdr file")
    digifil.add_argument("--fil_out_dir", type=str, default=None, help="directory for the filterbank; default: next to the vdif")
    digifil.add_argument("--nchan", type=int, default=512, help="channels per subband. Default=%(default)s.")
    digifil.add_argument("--nsec", type=float, default=120, help="seconds to process. Default=%(default)s.")
    digifil.add_argument("--start", type=float, default=1, help="seconds into the file to start at. Default=%(default)s.")
    digifil.add_argument("--force", action="store_true", help="delete a pre-existing filterbank of the same name")
    digifil.add_argument("--pol", type=int, default=2, choices=[0, 1, 2, 3, 4],
                         help="0/1: that polarisation; 2: Stokes I; 3: (PP+QQ)^2; 4: PP,QQ,PQ,QP. Default=%(default)s.")
    digifil.add_argument("--nbit", default=8, type=int, choices=VALID_NBIT, help="output bits (-32 = float). Default=%(default)s.")
    digifil.add_argument("--keepBP", action="store_true", help="do not rescale (digifil -I0): bandpass stays visible")
    digifil.add_argument("--tscrunch", type=int, default=1, help="downsampling factor (digifil -t). Default=%(default)s.")
    digifil.add_argument("--nthreads", type=int, default=1, help="kept for compatibility; the GPU path ignores it. Default=%(default)s.")
    prep.add_argument("--do_prepdata", action="store_true", help="run prepdata/prepsubband on the filterbank")
    prep.add_argument("--ncpus", type=int, default=1, help="1: prepdata, >1: prepsubband. Default=%(default)s.")
    prep.add_argument("--dm", type=float, default=None, help="dispersion measure; default from psrcat")
    prep.add_argument("--nozerodm", action="store_false", help="do not add -zerodm")
    prep.add_argument("--clip", type=int, default=5, help="clip S/N for prepdata/prepsubband, 0 = none. Default=%(default)s.")
    prep.add_argument("--dm2", type=float, default=0.0, help="upper DM of a prepsubband range. Default: single DM.")
    prep.add_argument("--dmstep", type=float, default=1.0, help="DM step. Default=%(default)s.")
    return parser.parse_args(argv)


def psr_info(psr):
    """ra, dec, dm from psrcat (process_vdif.py:102-108); psrcat stays an external tool."""
    query = "psrcat -c 'raj decj dm' -o short -nohead -nonumber {0}".format(psr)
    try:
        fields = subprocess.check_output(query, shell=True).split()
        ra, dec, dm = fields
    except Exception:
        raise RunError("psrcat died on given source {0}".format(psr))
    return ra.decode(), dec.decode(), float(dm)


def id_generator(size=20, chars=string.ascii_uppercase + string.digits + string.ascii_lowercase):
    return "".join(random.choice(chars) for _ in range(size))


_HDR_KEYS = ("TELESCOPE", "SOURCE", "RA", "DEC", "FREQ", "BW", "DATAFILE")


def make_hdr(psr, freq, filename, pol=2, usb=True, ra=None, dec=None, bw=16.0, telescope="ONSALA85", npol=2,
             tmp=False):
    """Write the 12-line ASCII side file (no trailing newline) that tells the channeliser how to
    read the raw file; negative BW marks LSB (process_vdif.py:115-139)."""
    signed_bw = bw if usb else -bw
    where = "/tmp/" if tmp else os.path.dirname(filename)
    if ra is None or dec is None:
        ra, dec, _ = psr_info(psr)
    values = dict(zip(_HDR_KEYS, (telescope, psr, ra, dec, freq, signed_bw, filename)))
    lines = ["HDR_VERSION 0.1"]
    lines += ["{0:<10} {1}".format(k, values[k]) for k in _HDR_KEYS]
    lines += ["INSTRUMENT VDIF", "MODE       PSR", "BASIS      Circular", "NPOL       {0}".format(npol)]
    hdrfile = "{0}/{1}_pol{2}.hdr".format(where, os.path.basename(filename), pol)
    with open(hdrfile, "w") as f:
        f.write("\n".join(lines))
    return hdrfile


def digifil_command(hdr, filterbankfile, start, nsecs, nchan, pol, nbit, tscrunch, nthreads, dm, coherent, keepBP,
                    iquv=False):
    """The digifil command line of process_vdif.py:156-182, token for token (``iquv`` appends the extension flag)."""
    if pol not in (0, 1, 2, 3, 4):
        raise InputError(f"pol = {pol} not implemented. Choices are 0, 1, 2, 3, 4")
    words = ["digifil", "-cont", "-c", f"-b{nbit}", f"-S{start}", f"-T{nsecs}", "-2", "-D", "0.0"]
    if tscrunch > 1:
        words += ["-t", str(tscrunch)]
    words += ["-o", filterbankfile, hdr, "-threads", str(nthreads)]
    leakage_factor = 512 if nchan <= 128 else 2 * nchan
    words.append(f"-P{pol}" if pol < 2 else {2: "-d1", 3: "-d3", 4: "-d4"}[pol])
    words.append(f"-F{nchan}:{leakage_factor}")
    if dm > 0.0:
        words += ["-D", str(dm)]
        if coherent:
            words.append(f"-F{nchan}:D")
    if keepBP:
        words.append("-I0")
    if iquv:
        if pol != 4:
            raise InputError("iquv needs pol = 4 (the four products)")
        words.append("-iquv")
    return " ".join(words)


def run_digifil(hdr, fil_out_dir=None, start=1, nsecs=120, nchan=128, overwrite=False, pol=2, nbit=8, tscrunch=1,
                nthreads=1, dm=0.0, coherent=False, keepBP=False, backend="abi", device=None, iquv=False):
    """Channelise the VDIF named by ``hdr`` into ``<fil_out_dir>/<hdr basename>.fil``
    (process_vdif.py:142-199).  A pre-existing FIFO at that path is written into, never removed."""
    filterbankfile = hdr.replace(".hdr", ".fil")
    if fil_out_dir is not None:
        filterbankfile = "{0}/{1}".format(fil_out_dir, os.path.basename(filterbankfile))
    if os.path.exists(filterbankfile):
        if not overwrite:
            raise InputError("Filterbankfile {0} exists already. ".format(filterbankfile) +
                             "Delete first or set --force to overwrite")
        if not stat.S_ISFIFO(os.stat(filterbankfile).st_mode):
            os.remove(filterbankfile)
    if nbit not in VALID_NBIT:
        raise InputError(f"nbit={nbit} not in supported values of {VALID_NBIT}. ")
    cmd = digifil_command(hdr, filterbankfile, start, nsecs, nchan, pol, nbit, tscrunch, nthreads, dm, coherent, keepBP,
                          iquv=iquv)
    print("running {0}".format(cmd))
    if backend == "shim":
        _run_shim(cmd, device)
    else:
        _run_abi(cmd, device)
    return filterbankfile


def _run_abi(cmd, device):
    from . import digifil_args
    try:
        cfg, _hdr, out = digifil_args.parse(cmd)
        if device is not None:
            cfg.device = int(device)
        with Channeliser(cfg) as chan:
            chan.run_file(cfg.datafile.decode(), out)
    except InputError as exc:  # a digifil that rejects its arguments dies -> RunError upstream
        raise RunError(f"Digifil died. \n stdout reports \n [] \n stderr reports \n [{exc.message!r}]")
    except RunError as exc:
        raise RunError(f"Digifil died. \n stdout reports \n [] \n stderr reports \n [{exc.message!r}]")


def _run_shim(cmd, device):
    errfile_nme = "/tmp/digifil.{0}".format(id_generator())
    outfile_nme = "/tmp/digifil.{0}".format(id_generator())
    env = dict(os.environ)
    env["PATH"] = os.path.dirname(SHIM) + os.pathsep + env.get("PATH", "")
    if device is not None:
        env["FRBCH_DEVICE"] = str(int(device))
    try:
        with open(errfile_nme, "w") as errfile, open(outfile_nme, "w") as outfile:
            subprocess.check_call(cmd, shell=True, stdout=outfile, stderr=errfile, env=env)
    except subprocess.CalledProcessError:
        with open(outfile_nme, "r") as f:
            stdout = f.readlines()
        with open(errfile_nme, "r") as f:
            stderr = f.readlines()
        raise RunError(f"Digifil died. \n stdout reports \n {stdout} \n stderr reports \n {stderr}")
    finally:
        for nme in (errfile_nme, outfile_nme):
            if os.path.exists(nme):
                os.remove(nme)


def prepdata(filterbankfile, dm1, zerodm=True, clip=5, dm2=0, dmstep=1.0, ncpus=1):
    """PRESTO prepdata / prepsubband on the filterbank (process_vdif.py:202-229); PRESTO stays an
    external tool, this only assembles and launches the command."""
    if dm2 > 0.0:
        if dm2 < dm1:
            raise InputError("DM2 must be larger than DM1.")
        numdms = int((dm2 - dm1) // dmstep + 1)
        cmd = "prepsubband -lodm {0} -numdms {1} -dmstep {2}".format(dm1, numdms, dmstep)
        outfile = filterbankfile.replace(".fil", "")
    else:
        cmd = "prepdata -dm {0}".format(dm1)
        outfile = filterbankfile.replace(".fil", "_dm{0}".format(dm1))
    cmd += " -filterbank -noweights -noscales -nobary -ncpus {0}".format(ncpus)
    if zerodm:
        cmd += " -zerodm "
    if clip > 0:
        cmd += " -clip {0} ".format(clip)
    cmd += " -o {0} {1}".format(outfile, filterbankfile)
    print("running {0}".format(cmd))
    try:
        subprocess.check_call(cmd, shell=True)
    except subprocess.CalledProcessError:
        raise RunError("Prepdata died.")


def main(argv=None):
    args = options(argv)
    if not args.usb and not args.lsb:
        raise InputError("You MUST supply either -l OR -u to specify if data are LSB or USB")
    if args.usb and args.lsb:
        raise InputError("You MUST supply either -l OR -u not both.")
    hdr = make_hdr(args.psrname, args.freq, args.filename, usb=bool(args.usb), bw=args.bw, telescope=args.telescope,
                   tmp=args.use_tmp, ra=args.ra, dec=args.dec, pol=args.pol)
    if args.hdr_only:
        print("Not creating filterbanks. Hdr files done.")
        return 0
    # one process named "digifil" per IF, as base2fil.sh's pwait throttle counts them (base2fil.sh:21-28)
    backend = os.environ.get("FRBCH_BACKEND", "shim")
    device = os.environ.get("FRBCH_DEVICE")
    iquv = os.environ.get("FRBCH_IQUV", "0") not in ("", "0")
    filterbankfile = run_digifil(hdr, args.fil_out_dir, args.start, args.nsec, args.nchan, overwrite=args.force,
                                 pol=args.pol, nbit=args.nbit, tscrunch=args.tscrunch, nthreads=args.nthreads,
                                 keepBP=args.keepBP, backend=backend, device=device, iquv=iquv)
    if args.do_prepdata:
        dm1 = args.dm if args.dm is not None else psr_info(args.psrname)[2]
        if os.environ.get("FRBCH_PREP", "presto") == "gpu":      # same options, dedispersed by libfrbch instead of PRESTO
            from . import post
            post.prepdata_gpu(filterbankfile, dm1, zerodm=args.nozerodm, clip=args.clip, dm2=args.dm2, dmstep=args.dmstep,
                              ncpus=args.ncpus, device=int(device or 0))
        else:
            prepdata(filterbankfile, dm1, zerodm=args.nozerodm, clip=args.clip, dm2=args.dm2, dmstep=args.dmstep,
                     ncpus=args.ncpus)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
