"""GPU parity tests proper: the HIP path, called through the C ABI, against the fp64 oracle."""
import numpy as np
import pytest

from tests import parity_util as pu

pytestmark = pytest.mark.gpu

CASES = [
    # bw, nchan, secs, kwargs
    (16.0, 128, 0.05, {}),                                   # BASELINE config 1 shape (-F128:512 -d1)
    (-16.0, 128, 0.05, {}),                                  # LSB: no band flip
    (16.0, 64, 0.05, dict(pol=4)),                           # -d4 coherency products
    (-16.0, 32, 0.05, dict(pol=4, nbit=-32, tscr=4, freq_res=64)),
    (16.0, 32, 0.05, dict(pol=0, nbit=2, tscr=2, freq_res=64)),
    (16.0, 32, 0.05, dict(pol=3, nbit=16, freq_res=64, interval=0.0)),   # -I0 (keepBP)
    (16.0, 32, 0.05, dict(pol=1, freq_res=64, interval=0.004, const=0)),  # no -c: per-interval rescale
    (16.0, 32, 0.05, dict(freq_res=64, interval=0.004, const=1, maxb=3)),
    (32.0, 1024, 0.14, {}),                                  # BASELINE config 2 shape, 2 blocks
]


@pytest.mark.parametrize("bw,nchan,secs,kw", CASES)
def test_fil_matches_oracle(hip_lib, bw, nchan, secs, kw):
    pu.run_streaming_case(hip_lib, bw, nchan, secs, **kw)
