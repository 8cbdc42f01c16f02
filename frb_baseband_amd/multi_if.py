"""Multi-IF orchestration: IF -> GPU/rank sharding and the host-side frequency concatenation.

The reference runs one process per IF (base2fil.sh:60-66), LSB (odd) IFs then USB (even) IFs
(:407-414), each writing a named FIFO, and joins them with sigproc ``splice`` whose argument list
is built highest IF first (base2fil.sh:350,367,422).  IFs never exchange data, so on a multi-GPU
node every rank owns whole IFs (IF i -> rank i mod world) and there is NO collective on the data
path; the only exchange is this host-side concatenation of [t][chan] rows.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

from . import sigproc


@dataclass
class IfPlan:
    index: int          # 1-based IF number as in <exp>_<st>_no0<scan>_IF<i>.vdif
    sideband: str       # 'l' (odd IFs) or 'u' (even IFs)   base2fil.sh:266-268,407-414
    freq_mhz: float     # value passed as -f: freqLSB_0 + (i-1)*bw  (base2fil.sh:54,65,254)


def plan_ifs(nif: int, freq_lsb_0: float, bw: float):
    """Frequency / sideband bookkeeping of run_process_vdif (base2fil.sh:30-67): within a sideband
    the centre steps by 2*bw; the USB series starts at freqLSB_0 + bw."""
    out = []
    for i in range(1, nif + 1):
        if i % 2 == 1:
            out.append(IfPlan(i, "l", freq_lsb_0 + ((i - 1) // 2) * 2.0 * bw))
        else:
            out.append(IfPlan(i, "u", (freq_lsb_0 + bw) + ((i - 2) // 2) * 2.0 * bw))
    return out


def shard(nif: int, world: int, rank: int):
    """IF numbers (1-based) owned by ``rank``: IF i -> rank (i-1) mod world."""
    return [i for i in range(1, nif + 1) if (i - 1) % world == rank]


def splice_order(nif: int):
    """Order in which base2fil.sh hands the per-IF filterbanks to splice: highest IF first."""
    return list(range(nif, 0, -1))


def _put_str(s: str) -> bytes:
    b = s.encode("ascii")
    return struct.pack("<i", len(b)) + b


def _header_bytes(h: dict) -> bytes:
    out = _put_str("HEADER_START")
    for key, val in h.items():
        out += _put_str(key)
        if isinstance(val, str):
            out += _put_str(val)
        elif isinstance(val, int):
            out += struct.pack("<i", val)
        else:
            out += struct.pack("<d", float(val))
    return out + _put_str("HEADER_END")


def _open_fil(x):
    """(header dict, header length, total bytes, reader(offset, n) -> bytes) of a path or a bytes object"""
    if isinstance(x, (bytes, bytearray)):
        buf = bytes(x)
        hdr, pos = sigproc.parse_header(buf)
        return hdr, pos, len(buf), (lambda off, n: buf[off:off + n]), None
    f = open(x, "rb")
    head = f.read(4096)
    hdr, pos = sigproc.parse_header(head)
    size = os.fstat(f.fileno()).st_size

    def rd(off, n):
        f.seek(off)
        return f.read(n)
    return hdr, pos, size, rd, f


def splice(inputs, out_path: str | None = None, block_rows: int = 4096):
    """Frequency-concatenate per-IF filterbanks given in DESCENDING frequency order (the order of
    base2fil.sh's splice_list).  ``inputs``: paths or bytes.  Rows are cut to the shortest input.
    Output: one SIGPROC file, nchans = sum, fch1 of the first input, [t][product][IF0 chans, IF1 ...].
    Streams ``block_rows`` rows at a time (a scan of 16 IFs x minutes never sits in memory); with ``out_path`` the file
    is written and its path returned, without it the bytes are returned (tests)."""
    srcs = [_open_fil(x) for x in inputs]
    try:
        first = srcs[0][0]
        for hdr, *_ in srcs[1:]:
            for key in ("nbits", "nifs", "tsamp"):
                if hdr[key] != first[key]:
                    raise ValueError(f"cannot splice: {key} differs ({hdr[key]} vs {first[key]})")
            if abs(hdr["tstart"] - first["tstart"]) > 0.5 * first["tsamp"] / 86400.0:
                raise ValueError("cannot splice: tstart differs")
        nbits, nifs = first["nbits"], first.get("nifs", 1)
        # bytes of one product line of each input (2-bit: 4 channels per byte)
        seg = [hdr["nchans"] * nbits // 8 for hdr, *_ in srcs]
        if nbits == 2 and any(hdr["nchans"] % 4 for hdr, *_ in srcs):
            raise ValueError("cannot splice 2-bit files whose channel count is not a multiple of 4")
        nt = min((size - pos) // (s_ * nifs) for (hdr, pos, size, rd, f), s_ in zip(srcs, seg))
        out_hdr = dict(first)
        out_hdr["nchans"] = int(sum(hdr["nchans"] for hdr, *_ in srcs))
        head = _header_bytes(out_hdr)
        sink = open(out_path, "wb") if out_path else None
        chunks = [head]
        if sink:
            sink.write(head)
        for r0 in range(0, nt, block_rows):
            nr = min(block_rows, nt - r0)
            parts = []
            for (hdr, pos, size, rd, f), s_ in zip(srcs, seg):
                raw = rd(pos + r0 * s_ * nifs, nr * s_ * nifs)
                parts.append(np.frombuffer(raw, dtype=np.uint8).reshape(nr, nifs, s_))
            block = np.concatenate(parts, axis=2).tobytes()      # byte-wise: whole bytes per (row, product, IF)
            if sink:
                sink.write(block)
            else:
                chunks.append(block)
        if sink:
            sink.close()
            return out_path
        return b"".join(chunks)
    finally:
        for *_, f in srcs:
            if f is not None:
                f.close()


def ifall_name(experiment: str, st: str, scanname: str, pol: int) -> str:
    """Name of the spliced product (base2fil.sh:389)."""
    return f"{experiment}_{st}_no0{scanname}_IFall_vdif_pol{pol}.fil"


def run_scan(channelisers, vdif_paths, out_fil: str) -> None:
    """One GPU, several IFs, one output: ``frbch_run_scan`` (include/frbch.h).  ``channelisers`` are freshly opened
    ``Channeliser`` objects in splice order (highest IF first, base2fil.sh:350,367); the frequency concatenation
    happens in device memory and a single IFall filterbank is written (no per-IF files, FIFOs or splice)."""
    import ctypes as C
    from . import channeliser as ch
    n = len(channelisers)
    assert n == len(vdif_paths) and n > 0
    lib = channelisers[0].lib
    handles = (C.c_void_p * n)(*[c._h for c in channelisers])
    paths = (C.c_char_p * n)(*[os.fsencode(p) for p in vdif_paths])
    rc = lib.frbch_run_scan(handles, n, paths, os.fsencode(out_fil))
    if rc < 0:
        raise (ch.InputError if rc == -1 else ch.RunError)(channelisers[0]._err(rc))


def scan_device(channelisers, d_frames_ptrs, nframes: int, frame_bytes: int, header_bytes: int, payload_off: int,
                nblocks: int, d_rows_ptr: int, rows_cap: int, flush: bool = True, stream: int = 0) -> int:
    """The same with everything resident in HBM: ``frbch_scan_device`` (include/frbch.h).  ``d_frames_ptrs[i]`` is the
    device address of the frames of ``channelisers[i]`` (splice order); the rows of all IFs land in ONE row buffer
    ``d_rows[row][product][IF-major channels]`` (row pitch = n x row_bytes of one IF).  Returns the rows written."""
    import ctypes as C
    from . import channeliser as ch
    n = len(channelisers)
    assert n == len(d_frames_ptrs) and n > 0
    lib = channelisers[0].lib
    handles = (C.c_void_p * n)(*[c._h for c in channelisers])
    frames = (C.c_void_p * n)(*[int(x) for x in d_frames_ptrs])
    rows = C.c_uint64(0)
    pitch = n * channelisers[0].info.row_bytes
    rc = lib.frbch_scan_device(handles, n, frames, nframes, frame_bytes, header_bytes, payload_off, nblocks,
                               1 if flush else 0, d_rows_ptr, pitch, rows_cap, C.byref(rows), stream or None)
    if rc < 0:
        raise (ch.InputError if rc == -1 else ch.RunError)(channelisers[0]._err(rc))
    return rows.value


def process_scan(vdif_by_if: dict, *, freq_lsb_0: float, bw: float, nchan: int, nsec: float, start: float = 0.0,
                 pol: int = 2, nbit: int = 8, tscrunch: int = 1, keepBP: bool = False, source: str = "unknown",
                 ra: str = "00:00:00.0", dec: str = "00:00:00.0", telescope: str = "ONSALA85",
                 out_dir: str, out_name: str = "IFall.fil", rank: int = 0, world: int = 1, local_device: int = 0,
                 barrier=None, lib=None, direct: bool = False):
    """Channelise this rank's IFs (one handle per IF on GPU ``local_device``) into
    ``out_dir/<vdif basename>_pol<pol>.fil`` and, on rank 0 after ``barrier()``, splice all IFs.

    ``vdif_by_if``: {IF number: path}.  No data-path collective: ranks only meet at the barrier.
    ``pol`` as run_digifil's, plus 5 = Stokes I,Q,U,V (the `-d4 -iquv` extension).
    ``direct=True``: the rank's IFs are concatenated on its GPU (``run_scan``) into one ``<out_name>.rank<r>`` piece
    and rank 0 splices the world's pieces (with one rank: the IFall file is written directly, no per-IF files).
    Returns the spliced path on rank 0, else None."""
    from . import channeliser as ch
    from . import process_vdif as pv
    nif = len(vdif_by_if)
    plans = {p.index: p for p in plan_ifs(nif, freq_lsb_0, bw)}
    mine = shard(nif, world, rank)
    if direct:
        from . import digifil_args
        # ranks own contiguous runs of the splice order so that their pieces concatenate without interleaving
        order = splice_order(nif)
        per = (nif + world - 1) // world
        mine = order[rank * per:(rank + 1) * per]
        chans, paths = [], []
        for i in mine:
            p = plans[i]
            hdr = pv.make_hdr(source, p.freq_mhz, vdif_by_if[i], pol=pol, usb=(p.sideband == "u"), ra=ra, dec=dec, bw=bw,
                              telescope=telescope)
            cmd = pv.digifil_command(hdr, os.path.join(out_dir, "unused.fil"), start, nsec, nchan, min(pol, 4), nbit, tscrunch,
                                     1, 0.0, False, keepBP, iquv=(pol == 5))
            cfg, _h, _o = digifil_args.parse(cmd, lib=lib)
            cfg.device = local_device
            chans.append(ch.Channeliser(cfg, lib))
            paths.append(vdif_by_if[i])
        out_path = os.path.join(out_dir, out_name)
        piece = out_path if world == 1 else f"{out_path}.rank{rank}"
        try:
            if chans:
                run_scan(chans, paths, piece)
        finally:
            for c in chans:
                c.close()
        if barrier is not None:
            barrier()
        if rank != 0:
            return None
        if world > 1:
            pieces = [f"{out_path}.rank{r}" for r in range(world) if order[r * per:(r + 1) * per]]
            splice(pieces, out_path)
        return out_path
    for i in mine:
        p = plans[i]
        path = vdif_by_if[i]
        hdr = pv.make_hdr(source, p.freq_mhz, path, pol=pol, usb=(p.sideband == "u"), ra=ra, dec=dec, bw=bw,
                          telescope=telescope)
        fil = os.path.join(out_dir, os.path.basename(hdr).replace(".hdr", ".fil"))
        cmd = pv.digifil_command(hdr, fil, start, nsec, nchan, min(pol, 4), nbit, tscrunch, 1, 0.0, False, keepBP,
                                 iquv=(pol == 5))
        from . import digifil_args
        cfg, _h, out = digifil_args.parse(cmd, lib=lib)
        cfg.device = local_device
        with ch.Channeliser(cfg, lib) as chan:
            chan.run_file(path, out)
    if barrier is not None:
        barrier()
    if rank != 0:
        return None
    ordered = []
    for i in splice_order(nif):
        base = os.path.basename(vdif_by_if[i]) + f"_pol{pol}.fil"
        ordered.append(os.path.join(out_dir, base))
    out_path = os.path.join(out_dir, out_name)
    splice(ordered, out_path)
    return out_path
