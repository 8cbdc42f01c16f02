"""Seeded synthetic per-IF VDIF: unit-variance Gaussian noise per
polarisation, quantised to 2 bits at +-0.9816 sigma, plus a weak tone at the centre of channel
floor(nchan/3); packed as spif2file would write the split file (spif2file.sh:178-186)."""
from __future__ import annotations

import numpy as np

from . import vdif

SEED_BASE = 0xF4B0
THRESH = 0.9816


def quantise_2bit(x: np.ndarray) -> np.ndarray:
    """float -> offset-binary states 0..3 (thresholds -t, 0, +t)."""
    return ((x >= -THRESH).astype(np.uint8) + (x >= 0.0).astype(np.uint8)
            + (x >= THRESH).astype(np.uint8))


def make_states(nsamp: int, *, if_index: int = 0, nchan: int = 1024, tone_amp: float = 0.1,
                sample0: int = 0, chunk: int = 0, impulse_at: int | None = None) -> np.ndarray:
    """u8[2][nsamp] states; the noise stream is seeded by (SEED_BASE + if_index, chunk)."""
    rng = np.random.default_rng([SEED_BASE + if_index, chunk])
    x = rng.standard_normal((2, nsamp), dtype=np.float32)
    if tone_amp:
        # tone at the centre of channel k: phase 2 pi (k + 1/2) n / (2 nchan) = 2 pi ((2k + 1) n mod 4 nchan) / (4 nchan) -- exact in
        # integers, so a table of 4 nchan entries replaces cos / sin of every sample (4 s per second of a 32 MHz IF before)
        k = nchan // 3
        period = 4 * nchan
        idx = ((2 * k + 1) * np.arange(sample0, sample0 + nsamp, dtype=np.int64)) % period
        tab = 2.0 * np.pi * np.arange(period, dtype=np.float64) / period
        x[0] += (tone_amp * np.sqrt(2.0) * np.cos(tab)).astype(np.float32)[idx]
        x[1] += (tone_amp * np.sqrt(2.0) * np.sin(tab)).astype(np.float32)[idx]
    if impulse_at is not None and sample0 <= impulse_at < sample0 + nsamp:
        x[:, impulse_at - sample0] = 10.0
    return quantise_2bit(x)


_CACHE: dict = {}          # the last few streams of a second or more (the test suites ask for the same 10-s IFs again and again)
_CACHE_MAX_BYTES = 3 << 30


def make_vdif(seconds: float, *, bw_mhz: float = 32.0, if_index: int = 0, nchan: int = 1024,
              tone_amp: float = 0.1, payload_bytes: int = vdif.DEFAULT_PAYLOAD, legacy: int = 0,
              seconds0: int = 1000, ref_epoch: int = 40, extra_frames: int = 0, bits: int = 2) -> np.ndarray:
    """Whole per-IF file as a uint8 array: ``seconds`` of data (+ ``extra_frames``; the reference's
    split adds 16, spif2file.sh:151), starting on a second boundary (spif2file.sh:148).  Streams of a second or more are kept
    (read-only) and returned again for the same arguments."""
    key = (float(seconds), float(bw_mhz), if_index, nchan, float(tone_amp), payload_bytes, legacy, seconds0, ref_epoch, extra_frames, bits)
    if key in _CACHE:
        return _CACHE[key]
    out = _make_vdif(seconds, bw_mhz=bw_mhz, if_index=if_index, nchan=nchan, tone_amp=tone_amp, payload_bytes=payload_bytes,
                     legacy=legacy, seconds0=seconds0, ref_epoch=ref_epoch, extra_frames=extra_frames, bits=bits)
    if seconds >= 1.0:
        out.flags.writeable = False
        while _CACHE and sum(v.nbytes for v in _CACHE.values()) + out.nbytes > _CACHE_MAX_BYTES:
            _CACHE.pop(next(iter(_CACHE)))
        _CACHE[key] = out
    return out


def _make_vdif(seconds, *, bw_mhz, if_index, nchan, tone_amp, payload_bytes, legacy, seconds0, ref_epoch, extra_frames, bits):
    fps = vdif.frames_per_second(bw_mhz, payload_bytes, bits)
    nfr = int(round(seconds * fps)) + extra_frames
    spf = payload_bytes * (4 // bits)            # dual-pol time samples per frame
    chunks = []
    per = max(1, (1 << 22) // spf) * spf         # ~4M samples per chunk, whole frames
    total = nfr * spf
    s = 0
    while s < total:
        n = min(per, total - s)
        st = make_states(n, if_index=if_index, nchan=nchan, tone_amp=tone_amp, sample0=s,
                         chunk=s // per)
        chunks.append(vdif.pack_states(st) if bits == 2 else vdif.pack_states_1bit((st >= 2).astype(np.uint8)))
        s += n
    payload = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return vdif.frame_payload(payload, bw_mhz=bw_mhz, seconds0=seconds0, ref_epoch=ref_epoch,
                              payload_bytes=payload_bytes, legacy=legacy, bits=bits)
