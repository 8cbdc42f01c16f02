"""After the filterbank: GPU incoherent dedispersion (the reference's `prepdata` stage) and GPU phase fold (the
reference's `dspsr -E <par> ... <IFall.fil>` stage) of SIGPROC filterbank files, through the C ABI (include/frbch.h).

* ``prepdata_gpu`` mirrors ``process_vdif.prepdata`` (process_vdif.py:202-229): same arguments, same output names
  (PRESTO's ``<outfile>.dat`` / ``.inf`` for one DM, ``<outfile>_DM<dm>.dat`` / ``.inf`` for a DM range), data
  dedispersed on the GPU instead of by PRESTO.  `-nobary -noweights -noscales` are what the reference always passes:
  topocentric, unweighted.
* ``fold_fil`` mirrors base2fil.sh:465-493: spin parameters from a psrcat-style .par file, 10-s sub-integrations,
  the filterbank's channels, plus the plot (PNG) of the dedispersed, time- and frequency-scrunched profile that
  ``psrplot -pF ... -j dedisperse,tscrunch,pscrunch,"fscrunch 128"`` draws.
There is no CPU fallback: the sums run in libfrbch.so on a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import zlib

import numpy as np

from . import _lib, sigproc
from .channeliser import InputError, RunError

DM_CONST = 1.0 / 2.41e-4      # s MHz^2 per pc cm^-3 (DSPSR / PRESTO)


def fil_desc(hdr: dict, product: int = 0) -> _lib.FrbchFilDesc:
    d = _lib.FrbchFilDesc()
    d.size = C.sizeof(_lib.FrbchFilDesc)
    d.nchan = hdr["nchans"]
    d.nifs = hdr.get("nifs", 1)
    d.nbits = hdr["nbits"]
    d.product = product
    d.fch1_mhz = hdr["fch1"]
    d.foff_mhz = hdr["foff"]
    d.tsamp_s = hdr["tsamp"]
    d.tstart_mjd = hdr["tstart"]
    return d


def _rows_of(fil: sigproc.SigprocFile) -> np.ndarray:
    if fil.header["nbits"] not in (8, 16, 32):
        raise InputError(f"nbits = {fil.header['nbits']}: the GPU stages take 8-, 16-bit or float32 filterbanks")
    return np.ascontiguousarray(fil.data)


def _check(rc: int, err) -> None:
    if rc < 0:
        msg = err.value.decode() if err is not None else ""
        raise (InputError if rc == _lib.E_ARG else RunError)(f"{_lib.load().frbch_strerror(rc).decode()}: {msg}")


def dm_list(dm1: float, dm2: float = 0.0, dmstep: float = 1.0):
    """the DMs of process_vdif.prepdata: one, or numdms = int((dm2-dm1)//dmstep + 1) from dm1 (process_vdif.py:209-214)"""
    if dm2 > 0.0:
        if dm2 < dm1:
            raise InputError("DM2 must be larger than DM1.")
        numdms = int((dm2 - dm1) // dmstep + 1)
        return [dm1 + i * dmstep for i in range(numdms)]
    return [dm1]


def dedisperse(fil: sigproc.SigprocFile, dms, zerodm: bool = True, clip: float = 5.0, device: int = 0, lib=None):
    """-> (float32 [ndm][nout], number of clipped time samples)"""
    lib = lib or _lib.load()
    rows = _rows_of(fil)
    desc = fil_desc(fil.header)
    dm_arr = np.ascontiguousarray(dms, dtype=np.float64)
    nout = lib.frbch_dedisperse_nout(C.byref(desc), rows.shape[0], dm_arr.ctypes.data, dm_arr.size)
    if nout <= 0:
        raise InputError("the dispersion delay across the band exceeds the length of the filterbank")
    out = np.empty((dm_arr.size, nout), dtype=np.float32)
    nclip = C.c_uint64(0)
    err = C.create_string_buffer(512)
    _check(lib.frbch_dedisperse_host(C.byref(desc), rows.ctypes.data, rows.shape[0], dm_arr.ctypes.data, dm_arr.size,
                                     1 if zerodm else 0, float(clip), device, out.ctypes.data, nout, C.byref(nclip),
                                     err, len(err)), err)
    return out, nclip.value


def write_inf(path: str, *, basename: str, hdr: dict, nsamp: int, dm: float, clipped: int) -> None:
    """PRESTO .inf side file (the keys `readfile` / `accelsearch` read), topocentric"""
    lo = hdr["fch1"] + (hdr["nchans"] - 1) * hdr["foff"] if hdr["foff"] < 0 else hdr["fch1"]
    bw = abs(hdr["foff"]) * hdr["nchans"]
    lines = [
        (" Data file name without suffix", basename),
        (" Telescope used", "Unknown"),
        (" Instrument used", "frbch (MI355X)"),
        (" Object being observed", hdr.get("source_name", "Unknown")),
        (" J2000 Right Ascension (hh:mm:ss.ssss)", _sex(hdr.get("src_raj", 0.0))),
        (" J2000 Declination     (dd:mm:ss.ssss)", _sex(hdr.get("src_dej", 0.0))),
        (" Data observed by", "unset"),
        (" Epoch of observation (MJD)", "%.15f" % hdr["tstart"]),
        (" Barycentered?           (1=yes, 0=no)", "0"),
        (" Number of bins in the time series", str(nsamp)),
        (" Width of each time series bin (sec)", "%.15g" % hdr["tsamp"]),
        (" Any breaks in the data? (1=yes, 0=no)", "0"),
        (" Type of observation (EM band)", "Radio"),
        (" Beam diameter (arcsec)", "0"),
        (" Dispersion measure (cm-3 pc)", "%.12g" % dm),
        (" Central freq of low channel (MHz)", "%.12g" % (lo - 0.0)),
        (" Total bandwidth (MHz)", "%.12g" % bw),
        (" Number of channels", str(hdr["nchans"])),
        (" Channel bandwidth (MHz)", "%.12g" % abs(hdr["foff"])),
        (" Data analyzed by", "frb_baseband_amd"),
        (" Any additional notes", "\n    GPU incoherent dedispersion, %d time samples clipped" % clipped),
    ]
    with open(path, "w") as f:
        for key, val in lines:
            f.write("%-40s=  %s\n" % (key, val))


def _sex(packed: float) -> str:
    sign = "-" if packed < 0 else ""
    p = abs(packed)
    hh = int(p // 10000)
    mm = int((p - hh * 10000) // 100)
    ss = p - hh * 10000 - mm * 100
    return "%s%02d:%02d:%07.4f" % (sign, hh, mm, ss)


def prepdata_gpu(filterbankfile, dm1, zerodm=True, clip=5, dm2=0, dmstep=1.0, ncpus=1, device=0, lib=None):
    """GPU replacement of process_vdif.prepdata (same signature; ``ncpus`` is accepted and ignored).  Returns the list of
    .dat files written."""
    fil = sigproc.read_fil(filterbankfile)
    dms = dm_list(dm1, dm2, dmstep)
    series, nclip = dedisperse(fil, dms, zerodm=zerodm, clip=float(clip), device=device, lib=lib)
    if dm2 > 0.0:
        base = filterbankfile.replace(".fil", "")
        names = ["%s_DM%.2f" % (base, dm) for dm in dms]          # prepsubband's naming
    else:
        names = [filterbankfile.replace(".fil", "_dm{0}".format(dm1))]   # process_vdif.py:216
    out = []
    for name, dm, y in zip(names, dms, series):
        y.astype("<f4").tofile(name + ".dat")
        write_inf(name + ".inf", basename=os.path.basename(name), hdr=fil.header, nsamp=y.size, dm=dm, clipped=nclip)
        out.append(name + ".dat")
    return out


# ------------------------------------------------------------------------------------------------------------------
# fold
# ------------------------------------------------------------------------------------------------------------------
def read_par(path: str) -> dict:
    """F0 / F1 / PEPOCH / DM of a psrcat -e style ephemeris (base2fil.sh:465: `psrcat -e <target> > <target>.psrcat.par`);
    P0 / P1 are accepted instead of F0 / F1."""
    vals = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 2:
                try:
                    vals[parts[0].upper()] = float(parts[1].replace("D", "E"))
                except ValueError:
                    vals[parts[0].upper()] = parts[1]
    if "F0" not in vals:
        if "P0" not in vals or not isinstance(vals["P0"], float):
            raise InputError(f"{path}: neither F0 nor P0")
        p0, p1 = vals["P0"], float(vals.get("P1", 0.0) or 0.0)
        vals["F0"] = 1.0 / p0
        vals["F1"] = -p1 / (p0 * p0)
    out = {"F0": float(vals["F0"]), "F1": float(vals.get("F1", 0.0) or 0.0), "DM": float(vals.get("DM", 0.0) or 0.0)}
    out["PEPOCH"] = float(vals["PEPOCH"]) if isinstance(vals.get("PEPOCH"), float) else None
    out["PSR"] = str(vals.get("PSRJ", vals.get("PSRB", vals.get("PSR", "unknown"))))
    return out


def default_nbin(f0: float, tsamp: float, cap: int = 1024) -> int:
    """largest power of two not above period / tsamp (no bin narrower than a sample), at most ``cap``"""
    n = max(2.0, 1.0 / (f0 * tsamp))
    nb = 2
    while nb * 2 <= n and nb * 2 <= cap:
        nb *= 2
    return nb


def fold(fil: sigproc.SigprocFile, par: dict, nbin: int = 0, subint_s: float = 10.0, apply_delays: bool = False,
         device: int = 0, lib=None):
    """-> (profile float64 [nsub][nchan][nbin] sums, hits uint32 same shape, nbin).  PEPOCH defaults to tstart."""
    lib = lib or _lib.load()
    rows = _rows_of(fil)
    desc = fil_desc(fil.header)
    nbin = nbin or default_nbin(par["F0"], fil.header["tsamp"])
    pepoch = par["PEPOCH"] if par.get("PEPOCH") is not None else fil.header["tstart"]
    nsub = lib.frbch_fold_nsub(C.byref(desc), rows.shape[0], float(subint_s))
    if nsub <= 0:
        raise InputError("bad sub-integration length")
    prof = np.zeros((nsub, nbin, desc.nchan), dtype=np.float64)
    hits = np.zeros((nsub, nbin, desc.nchan), dtype=np.uint32)
    err = C.create_string_buffer(512)
    _check(lib.frbch_fold_host(C.byref(desc), rows.ctypes.data, rows.shape[0], par["F0"], par["F1"], pepoch, par["DM"],
                               1 if apply_delays else 0, nbin, float(subint_s), device, prof.ctypes.data, hits.ctypes.data,
                               nsub, err, len(err)), err)
    return prof.transpose(0, 2, 1).copy(), hits.transpose(0, 2, 1).copy(), nbin


def dedisperse_profile(prof: np.ndarray, hdr: dict, f0: float, dm: float) -> np.ndarray:
    """rotate every channel's profile by its dispersion delay (nearest bin): `psrplot -j dedisperse`"""
    nsub, nchan, nbin = prof.shape
    fc = hdr["fch1"] + np.arange(nchan) * hdr["foff"]
    fhi = fc.max()
    delay = dm * DM_CONST * (fc ** -2 - fhi ** -2)
    shift = np.rint(delay * f0 * nbin).astype(np.int64) % nbin
    out = np.empty_like(prof)
    for c in range(nchan):
        out[:, c, :] = np.roll(prof[:, c, :], -int(shift[c]), axis=1)
    return out


ARCHIVE_MAGIC = b"FRBFOLD1"


def write_archive(path: str, prof: np.ndarray, hits: np.ndarray, meta: dict) -> None:
    """one file: magic, JSON header (length-prefixed), float64 sums [nsub][nchan][nbin], uint32 hits"""
    head = json.dumps(meta, sort_keys=True).encode()
    with open(path, "wb") as f:
        f.write(ARCHIVE_MAGIC + struct.pack("<I", len(head)) + head)
        f.write(np.ascontiguousarray(prof, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(hits, dtype="<u4").tobytes())


def read_archive(path: str):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != ARCHIVE_MAGIC:
        raise InputError(f"{path}: not a FRBFOLD1 archive")
    (n,) = struct.unpack_from("<I", buf, 8)
    meta = json.loads(buf[12:12 + n].decode())
    shape = (meta["nsub"], meta["nchan"], meta["nbin"])
    cnt = shape[0] * shape[1] * shape[2]
    prof = np.frombuffer(buf, dtype="<f8", count=cnt, offset=12 + n).reshape(shape)
    hits = np.frombuffer(buf, dtype="<u4", count=cnt, offset=12 + n + 8 * cnt).reshape(shape)
    return prof, hits, meta


def write_png(path: str, img: np.ndarray) -> None:
    """minimal 8-bit greyscale PNG writer (no plotting library in the pipeline image)"""
    a = np.asarray(img, dtype=np.float64)
    lo, hi = float(a.min()), float(a.max())
    g = np.zeros(a.shape, np.uint8) if hi <= lo else np.clip((a - lo) / (hi - lo) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    raw = b"".join(b"\x00" + g[r].tobytes() for r in range(g.shape[0]))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", g.shape[1], g.shape[0], 8, 0, 0, 0, 0))
    png += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def fold_fil(filterbankfile: str, parfile: str, nbin: int = 0, subint_s: float = 10.0, fscrunch_to: int = 128,
             device: int = 0, lib=None, out_base: str | None = None):
    """base2fil.sh:465-493 for one filterbank: fold with the ephemeris, write ``<fil>.ar`` (FRBFOLD1), the profile as
    text (``<fil>.profile.txt``: bin, mean flux of the dedispersed, scrunched profile) and the phase-frequency plot
    ``<fil>.png`` (dedispersed, time-scrunched, frequency-scrunched to ``fscrunch_to`` channels, profile strip on top)."""
    fil = sigproc.read_fil(filterbankfile)
    par = read_par(parfile)
    prof, hits, nbin = fold(fil, par, nbin=nbin, subint_s=subint_s, apply_delays=False, device=device, lib=lib)
    base = out_base or filterbankfile
    hdr = fil.header
    meta = {"source": par["PSR"], "f0": par["F0"], "f1": par["F1"], "dm": par["DM"],
            "pepoch": par["PEPOCH"] if par["PEPOCH"] is not None else hdr["tstart"], "dedispersed": False,
            "nsub": int(prof.shape[0]), "nchan": int(prof.shape[1]), "nbin": int(nbin), "subint_s": subint_s,
            "tstart": hdr["tstart"], "tsamp": hdr["tsamp"], "fch1": hdr["fch1"], "foff": hdr["foff"],
            "timing": "topocentric polynomial F0, F1 about PEPOCH (no barycentric, binary or position terms)"}
    write_archive(base + ".ar", prof, hits, meta)
    dd = dedisperse_profile(prof, hdr, par["F0"], par["DM"])
    hh = dedisperse_profile(hits.astype(np.float64), hdr, par["F0"], par["DM"])
    tot, cnt = dd.sum(axis=0), hh.sum(axis=0)                                   # tscrunch
    nchan = tot.shape[0]
    fs = max(1, nchan // max(1, min(fscrunch_to, nchan)))
    nf = nchan // fs
    tot_f = tot[: nf * fs].reshape(nf, fs, nbin).sum(axis=1)
    cnt_f = cnt[: nf * fs].reshape(nf, fs, nbin).sum(axis=1)
    mean_f = np.where(cnt_f > 0, tot_f / np.maximum(cnt_f, 1), 0.0)
    profile = np.where(cnt.sum(axis=0) > 0, tot.sum(axis=0) / np.maximum(cnt.sum(axis=0), 1), 0.0)
    with open(base + ".profile.txt", "w") as f:
        f.write("# bin  mean_sample   (dedispersed DM=%g, tscrunched, fscrunched; %s)\n" % (par["DM"], par["PSR"]))
        for i, v in enumerate(profile):
            f.write("%d %.9g\n" % (i, v))
    strip = np.tile((profile - profile.min()) / max(1e-30, profile.max() - profile.min()), (max(8, nf // 8), 1))
    wf = mean_f - mean_f.mean(axis=1, keepdims=True)
    wf = (wf - wf.min()) / max(1e-30, wf.max() - wf.min())
    write_png(base + ".png", np.concatenate([strip, wf], axis=0))
    return base + ".ar", profile


def main(argv=None):
    """``python -m frb_baseband_amd.post fold <fil> <par> [...]`` / ``... prepdata <fil> --dm <dm> [...]``: the two stages
    as commands, for the places where base2fil.sh / process_vdif.py launch dspsr and prepdata"""
    import argparse
    ap = argparse.ArgumentParser(prog="frb_baseband_amd.post")
    sub = ap.add_subparsers(dest="cmd", required=True)
    f = sub.add_parser("fold", help="fold a filterbank with a .par file (base2fil.sh:465-493)")
    f.add_argument("fil")
    f.add_argument("par")
    f.add_argument("--nbin", type=int, default=0, help="phase bins (0: largest power of two <= period / tsamp, at most 1024)")
    f.add_argument("-L", "--subint", type=float, default=10.0, help="sub-integration length, s (dspsr -L 10)")
    f.add_argument("--fscrunch", type=int, default=128, help='channels of the plot (psrplot -j "fscrunch 128")')
    f.add_argument("--device", type=int, default=int(os.environ.get("FRBCH_DEVICE", "0")))
    d = sub.add_parser("prepdata", help="incoherent dedispersion (process_vdif.py:202-229)")
    d.add_argument("fil")
    d.add_argument("--dm", type=float, required=True)
    d.add_argument("--dm2", type=float, default=0.0)
    d.add_argument("--dmstep", type=float, default=1.0)
    d.add_argument("--nozerodm", action="store_false", help="do not subtract the zero-DM series")
    d.add_argument("--clip", type=float, default=5.0)
    d.add_argument("--device", type=int, default=int(os.environ.get("FRBCH_DEVICE", "0")))
    a = ap.parse_args(argv)
    if a.cmd == "fold":
        ar, profile = fold_fil(a.fil, a.par, nbin=a.nbin, subint_s=a.subint, fscrunch_to=a.fscrunch, device=a.device)
        print("wrote {0}, {1}.profile.txt, {1}.png; peak bin {2} of {3}".format(ar, a.fil, int(np.argmax(profile)), profile.size))
    else:
        for path in prepdata_gpu(a.fil, a.dm, zerodm=a.nozerodm, clip=a.clip, dm2=a.dm2, dmstep=a.dmstep, device=a.device):
            print("wrote", path)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
