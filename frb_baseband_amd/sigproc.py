"""SIGPROC filterbank reader (host side; used to inspect what the channeliser wrote and by the
frequency concatenation that replaces ``splice``, base2fil.sh:422-446)."""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

_INT_KEYS = {"telescope_id", "machine_id", "data_type", "barycentric", "pulsarcentric", "nbits",
             "nsamples", "nchans", "nifs", "nbeams", "ibeam"}
_STR_KEYS = {"rawdatafile", "source_name"}
_DBL_KEYS = {"az_start", "za_start", "src_raj", "src_dej", "tstart", "tsamp", "fch1", "foff",
             "refdm", "period"}


@dataclass
class SigprocFile:
    header: dict
    header_bytes: int
    data: np.ndarray  # [t][nifs][nchans]


def _rd_str(buf: bytes, pos: int):
    (n,) = struct.unpack_from("<i", buf, pos)
    if n < 0 or n > 80:
        raise ValueError(f"bad SIGPROC string length {n} at byte {pos}")
    return buf[pos + 4: pos + 4 + n].decode("ascii"), pos + 4 + n


def parse_header(buf: bytes):
    key, pos = _rd_str(buf, 0)
    if key != "HEADER_START":
        raise ValueError("not a SIGPROC filterbank (no HEADER_START)")
    hdr = {}
    while True:
        key, pos = _rd_str(buf, pos)
        if key == "HEADER_END":
            return hdr, pos
        if key in _INT_KEYS:
            (hdr[key],) = struct.unpack_from("<i", buf, pos)
            pos += 4
        elif key in _DBL_KEYS:
            (hdr[key],) = struct.unpack_from("<d", buf, pos)
            pos += 8
        elif key in _STR_KEYS:
            hdr[key], pos = _rd_str(buf, pos)
        else:
            raise ValueError(f"unknown SIGPROC header key {key!r}")


def unpack_samples(raw: bytes, nbits: int) -> np.ndarray:
    if nbits == 32:
        return np.frombuffer(raw, dtype="<f4")
    if nbits == 16:
        return np.frombuffer(raw, dtype="<u2")
    if nbits == 8:
        return np.frombuffer(raw, dtype=np.uint8)
    if nbits == 2:
        b = np.frombuffer(raw, dtype=np.uint8)
        out = np.empty(b.size * 4, dtype=np.uint8)
        for i in range(4):
            out[i::4] = (b >> (2 * i)) & 3
        return out
    raise ValueError(f"unsupported nbits {nbits}")


def read_fil(path_or_bytes) -> SigprocFile:
    if isinstance(path_or_bytes, (bytes, bytearray)):
        buf = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            buf = f.read()
    hdr, pos = parse_header(buf)
    flat = unpack_samples(buf[pos:], hdr["nbits"])
    row = hdr["nchans"] * hdr.get("nifs", 1)
    nt = flat.size // row
    data = flat[: nt * row].reshape(nt, hdr.get("nifs", 1), hdr["nchans"])
    return SigprocFile(header=hdr, header_bytes=pos, data=data)
