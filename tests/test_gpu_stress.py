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
    res, resc = [], []
    for flags in (0, 3):                                            # 3 = generic K1 + generic K2
        with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, secs, flags=flags, **kw), hip_lib) as c:
            res.append(sigproc.read_fil(c.channelise_bytes(raw)))
            resc.append(c.get_rescale())
            a_info_r = c.info.freq_res
    a, b = res
    assert a.header == b.header and a.data.shape == b.data.shape and a.data.shape[0] > 0
    x, y = a.data.astype(np.float64), b.data.astype(np.float64)
    # -b -32 writes x = (P + offset) * scale with offset = -mean(P), scale = 1 / sigma(P) per (product, channel).  Each path is
    # within power_rtol * (mean total power of the channel) of the exact P (parity_util), so the two differ by at most
    # twice that, times the scale applied, plus the fp32 rounding of x itself.
    off, sc = resc[1]                                               # [nif][C], input channel order
    pol = kw["pol"]
    mean_tot = -(off[0] + off[1]) if pol == 4 else -off[0]          # PP + QQ; I of IQUV; the product itself for one polarisation
    if pol in (0, 1):
        mean_tot = 2.0 * mean_tot
    if bw > 0:                                                      # rows are written in descending frequency
        mean_tot, sc = mean_tot[::-1], sc[:, ::-1]
    rtol = pu.power_rtol(nchan, a_info_r, kw["tscr"])
    slack = 1.0 if kw["interval"] >= 10.0 else 1.5                  # (the frozen scale of a short first interval)
    bound = 2.0 * slack * rtol * np.abs(mean_tot)[None, None, :] * np.abs(sc)[None, :, :] + 5e-7 * np.abs(y)
    worst = (np.abs(x - y) / bound).max()
    assert worst <= 1.0, (worst, kw)
