"""VDIF frame helpers for the per-IF files the channeliser consumes.

The per-IF file is what jive5ab's ``spif2file`` writes (spif2file.sh:178-186): single thread,
2 channels (= 2 polarisations) x 2 bit, real sampled, ``vdifsize`` payload bytes (8000 in every
recipe the reference uses, spif2file.sh:181) behind a 32-byte (16 if legacy) header.  Frame
geometry as consumed at base2fil.sh:395-401.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

DEFAULT_PAYLOAD = 8000


@dataclass
class VdifFrameHeader:
    seconds: int = 0
    ref_epoch: int = 0
    frame_nr: int = 0
    frame_bytes: int = 8032
    log2_nchan: int = 1
    bits_per_sample: int = 2
    thread_id: int = 0
    station_id: int = 0x4566  # 'Ef'
    legacy: int = 0
    invalid: int = 0
    is_complex: int = 0

    @property
    def header_bytes(self) -> int:
        return 16 if self.legacy else 32

    @property
    def payload_bytes(self) -> int:
        return self.frame_bytes - self.header_bytes

    def pack(self) -> bytes:
        w0 = (self.invalid << 31) | (self.legacy << 30) | (self.seconds & 0x3FFFFFFF)
        w1 = ((self.ref_epoch & 0x3F) << 24) | (self.frame_nr & 0xFFFFFF)
        w2 = ((self.log2_nchan & 0x1F) << 24) | ((self.frame_bytes // 8) & 0xFFFFFF)
        w3 = ((self.is_complex << 31) | (((self.bits_per_sample - 1) & 0x1F) << 26)
              | ((self.thread_id & 0x3FF) << 16) | (self.station_id & 0xFFFF))
        words = [w0, w1, w2, w3] if self.legacy else [w0, w1, w2, w3, 0, 0, 0, 0]
        return struct.pack("<%dI" % len(words), *words)

    @classmethod
    def unpack(cls, buf: bytes) -> "VdifFrameHeader":
        w0, w1, w2, w3 = struct.unpack_from("<4I", buf, 0)
        return cls(seconds=w0 & 0x3FFFFFFF, ref_epoch=(w1 >> 24) & 0x3F, frame_nr=w1 & 0xFFFFFF,
                   frame_bytes=(w2 & 0xFFFFFF) * 8, log2_nchan=(w2 >> 24) & 0x1F,
                   bits_per_sample=((w3 >> 26) & 0x1F) + 1, thread_id=(w3 >> 16) & 0x3FF,
                   station_id=w3 & 0xFFFF, legacy=(w0 >> 30) & 1, invalid=(w0 >> 31) & 1,
                   is_complex=(w3 >> 31) & 1)


def frames_per_second(bw_mhz: float, payload_bytes: int = DEFAULT_PAYLOAD, bits: int = 2) -> int:
    """2*|bw| Msamp/s x 2 pol x nbit / 8 / payload  (= bw*1e6/8000 for 2 bit and the standard payload;
    base2fil.sh:251,400-401)."""
    fps = abs(bw_mhz) * 1.0e6 * 2 * 2 * bits / 8 / payload_bytes
    if abs(fps - round(fps)) > 1e-9:
        raise ValueError(f"bw={bw_mhz} MHz does not give an integer frame rate for "
                         f"{payload_bytes}-byte payloads")
    return int(round(fps))


def pack_states(states: np.ndarray) -> np.ndarray:
    """u8[2][nsamp] offset-binary states 0..3 -> payload bytes u8[nsamp/2].

    bits[1:0]=pol0 t, [3:2]=pol1 t, [5:4]=pol0 t+1, [7:6]=pol1 t+1.
    """
    assert states.shape[0] == 2 and states.shape[1] % 2 == 0
    s = states.astype(np.uint8)
    return (s[0, 0::2] | (s[1, 0::2] << 2) | (s[0, 1::2] << 4) | (s[1, 1::2] << 6)).astype(np.uint8)


def pack_states_1bit(states: np.ndarray) -> np.ndarray:
    """u8[2][nsamp] states 0/1 -> payload bytes u8[nsamp/4]: bit 2i = pol0 sample i, bit 2i+1 = pol1 sample i."""
    assert states.shape[0] == 2 and states.shape[1] % 4 == 0
    s = states.astype(np.uint8)
    out = np.zeros(states.shape[1] // 4, dtype=np.uint8)
    for i in range(4):
        out |= (s[0, i::4] << (2 * i)) | (s[1, i::4] << (2 * i + 1))
    return out


def frame_payload(payload: np.ndarray, *, bw_mhz: float, seconds0: int = 0, ref_epoch: int = 40,
                  frame0: int = 0, payload_bytes: int = DEFAULT_PAYLOAD, legacy: int = 0,
                  station_id: int = 0x4566, bits: int = 2) -> np.ndarray:
    """Wrap a payload byte stream (whole frames) in VDIF headers -> u8 frame stream."""
    assert payload.size % payload_bytes == 0, "payload must be a whole number of frames"
    nfr = payload.size // payload_bytes
    fps = frames_per_second(bw_mhz, payload_bytes, bits)
    hb = 16 if legacy else 32
    out = np.empty((nfr, hb + payload_bytes), dtype=np.uint8)
    out[:, hb:] = payload.reshape(nfr, payload_bytes)
    idx = frame0 + np.arange(nfr, dtype=np.int64)
    secs = (seconds0 + idx // fps).astype(np.uint32)
    fnr = (idx % fps).astype(np.uint32)
    words = np.zeros((nfr, hb // 4), dtype="<u4")
    words[:, 0] = (np.uint32(legacy) << 30) | (secs & 0x3FFFFFFF)
    words[:, 1] = (np.uint32(ref_epoch & 0x3F) << 24) | (fnr & 0xFFFFFF)
    words[:, 2] = (np.uint32(1) << 24) | np.uint32((hb + payload_bytes) // 8)
    words[:, 3] = (np.uint32(bits - 1) << 26) | np.uint32(station_id & 0xFFFF)
    out[:, :hb] = words.view(np.uint8).reshape(nfr, hb)
    return out.reshape(-1)
