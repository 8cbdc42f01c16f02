"""Differential stress test (GPU): the register-pass kernels against the generic radix-2 kernels of the same library on
seeded random configurations -- channel count, tscrunch, products, band sense, batch size, rescale interval.  Both paths
are separately parity-tested against the oracle on hand-picked cases (test_gpu_parity.py); this test sweeps the
combinations those cases do not name (every tscrunch goes through some tile / two-stage / generic decision).  Float
output (-b -32), compared relative to the mean power: TEST INFRASTRUCTURE, no oracle involved."""
import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch, sigproc, synth
from tests import parity_util as pu

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20261004)
    out = []
    for i in range(28):
        nchan = int(rng.choice([128, 256, 512, 1024, 2048, 4096]))
        bw = float(rng.choice([32.0, 64.0] if nchan >= 2048 else [16.0, 32.0, 64.0])) * (1 if rng.random() < 0.5 else -1)
        n_block = 2 * nchan * 2 * nchan                              # samples per filterbank block (freq_res = 2 nchan)
        rate = 2.0 * abs(bw) * 1e6
        nblk = 2 if nchan >= 2048 else int(rng.integers(3, 9))
        secs = (nblk + 0.3) * n_block / rate
        tmax = min(64, 2 * nchan)
        tscr = int(2 ** rng.integers(0, int(np.log2(tmax)) + 1))
        pol = int(rng.choice([0, 1, 2, 2, 2, 4, 5]))
        interval = float(rng.choice([10.0, 10.0, secs * 0.45]))
        maxb = int(rng.choice([0, 0, 1, 2]))
        out.append((bw, nchan, round(secs, 6), dict(pol=pol, tscr=tscr, nbit=-32, interval=interval, maxb=maxb)))
    return out


@pytest.mark.parametrize("bw,nchan,secs,kw", _cases())
def test_fast_kernels_equal_generic_kernels(hip_lib, bw, nchan, secs, kw):
    raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan)
    res = []
    for flags in (0, 3):                                            # 3 = generic K1 + generic K2
        with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, secs, flags=flags, **kw), hip_lib) as c:
            res.append(sigproc.read_fil(c.channelise_bytes(raw)))
    a, b = res
    assert a.header == b.header and a.data.shape == b.data.shape and a.data.shape[0] > 0
    x, y = a.data.astype(np.float64), b.data.astype(np.float64)
    # -b -32 writes (P + offset) * scale: of order one per channel, so compare against that scale
    err = np.abs(x - y).max() / max(1.0, np.abs(y).std())
    assert err <= 2e-3, (err, kw)
