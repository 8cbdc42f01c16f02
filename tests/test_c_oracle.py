"""The plain-C fp32 oracle port (CPU baseline of bench.py) is pinned to the numpy fp64 oracle."""
import numpy as np
import pytest

from frb_baseband_amd import synth
from oracle import c_oracle
from oracle import frb_oracle as o


@pytest.mark.parametrize("c,r,pol,t", [(16, 64, 4, 2), (16, 64, 5, 1), (128, 512, 2, 1), (32, 64, 3, 4), (64, 512, 0, 8), (8, 16, 1, 1)])
def test_c_port_matches_numpy_oracle(c, r, pol, t):
    n = 2 * c * r
    raw = synth.make_vdif(3 * n / 32e6 + 0.001, bw_mhz=16.0, nchan=c)
    payload = o.strip_frames(raw, 8032, 32)
    got = c_oracle.block_power(payload, c, r, 3, pol, t)
    x = o.unpack_2bit(payload[: 3 * n // 2])
    ref = np.concatenate([o.tscrunch(o.detect(o.filterbank_block(x[:, b * n:(b + 1) * n], c, r), pol), t)
                          for b in range(3)], axis=2)
    scale = np.abs(ref[0]).mean()
    assert np.abs(got - ref).max() <= 2e-5 * scale


def test_c_unpack_all_bytes():
    import ctypes as C
    lib = c_oracle._load()
    b = np.arange(256, dtype=np.uint8)
    p0 = np.empty(512, np.float32)
    p1 = np.empty(512, np.float32)
    lib.frbo_unpack_2bit.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.frbo_unpack_2bit(b.ctypes.data, 256, p0.ctypes.data, p1.ctypes.data)
    x = o.unpack_2bit(b)
    assert np.array_equal(p0, x[0].astype(np.float32)) and np.array_equal(p1, x[1].astype(np.float32))
