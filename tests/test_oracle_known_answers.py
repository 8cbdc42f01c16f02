"""Known-answer tests that pin the CPU oracle (the reference ships none).
The numerical stages are 'parity unpinned' against digifil itself (not available); these tests pin
the oracle to the analytic behaviour of the stage chain the reference selects with its flags."""
import numpy as np
import pytest

from frb_baseband_amd import sigproc, synth, vdif
from oracle import frb_oracle as o


def test_unpack_all_byte_values():
    b = np.arange(256, dtype=np.uint8)
    x = o.unpack_2bit(b)                                   # [2][512]
    lv = o.LEVELS_2BIT
    for v in range(256):
        assert x[0, 2 * v] == lv[v & 3] and x[1, 2 * v] == lv[(v >> 2) & 3]
        assert x[0, 2 * v + 1] == lv[(v >> 4) & 3] and x[1, 2 * v + 1] == lv[(v >> 6) & 3]
    # pack_states is the inverse
    st = np.vstack([np.tile([0, 1, 2, 3], 4), np.repeat([0, 1, 2, 3], 4)]).astype(np.uint8)
    y = o.unpack_2bit(vdif.pack_states(st))
    assert np.array_equal(y[0], lv[st[0]]) and np.array_equal(y[1], lv[st[1]])


@pytest.mark.parametrize("k", [0, 5, 15])
def test_tone_lands_in_its_channel(k):
    c, r = 16, 64
    n = 2 * c * r
    t = np.arange(n)
    x = np.cos(2 * np.pi * ((k + 0.5) / (2.0 * c)) * t + 0.3)
    p = o.detect(o.filterbank_block(np.vstack([x, np.zeros(n)]), c, r), 2)[0]
    prof = p.mean(axis=1)
    assert prof.argmax() == k
    assert prof[k] / (prof.sum() - prof[k] + 1e-300) > 1e20     # exact bin: leakage is round-off only


def test_sideband_order_in_file():
    """USB input is flipped so that foff < 0; LSB input is written as is."""
    c, r = 16, 64
    n = 2 * c * r
    k = 3
    t = np.arange(2 * n)
    tone = np.cos(2 * np.pi * ((k + 0.5) / (2.0 * c)) * t)
    st = synth.quantise_2bit(np.vstack([tone * 2, tone * 2]))
    payload = vdif.pack_states(st)
    pad = (-payload.size) % 8000
    raw = vdif.frame_payload(np.concatenate([payload, np.zeros(pad, np.uint8) + 0x55]), bw_mhz=16.0)
    for bw, want in ((16.0, c - 1 - k), (-16.0, k)):
        cfg = o.Config(bw_mhz=bw, nchan=c, freq_res=r, total_s=2 * n / 32e6, nbit=-32, rescale_interval_s=0.0)
        f = sigproc.read_fil(o.channelise(raw, cfg))
        assert f.header["foff"] < 0
        assert f.header["fch1"] == pytest.approx(cfg.freq_mhz + 8.0 - 0.5)
        assert f.data[:, 0, :].mean(axis=0).argmax() == want


def test_impulse_is_flat_and_parseval():
    c, r = 8, 32
    n = 2 * c * r
    x = np.zeros((2, n))
    x[0, 37] = 1.0
    y = o.filterbank_block(x, c, r)
    p = o.detect(y, 0)[0]
    # every channel gets the same total power; Parseval for the unnormalised transforms:
    # sum_t |y_k[t]|^2 = R * sum_j |X[kR+j]|^2 = R * R (|X| = 1 for an impulse)
    np.testing.assert_allclose(p.sum(axis=1), np.full(c, float(r * r)), rtol=1e-12)


def test_noise_statistics_and_digitiser_levels():
    raw = synth.make_vdif(0.05, bw_mhz=16.0, nchan=64, tone_amp=0.0)
    cfg = o.Config(bw_mhz=16.0, nchan=64, total_s=0.05, pol_mode=2)
    f = sigproc.read_fil(o.channelise(raw, cfg))
    p = cfg.result["power"][0]
    ratio = p.mean(axis=1) ** 2 / p.var(axis=1)          # chi^2 with 4 dof: mean^2/var = 2
    assert np.median(ratio) == pytest.approx(2.0, rel=0.1)
    d = f.data.astype(np.float64)
    assert d.mean() == pytest.approx(127.5, abs=0.6)     # digi mean
    assert d.std() == pytest.approx(127.5 / 6.0, rel=0.06)  # digi scale x unit sigma (clipped tail)


def test_resolution_identity():
    """create_config.py:561: df = bw/nchan MHz, dt = nchan*tscrunch/bw us; rows = whole blocks only."""
    bw, c, t = 16.0, 32, 4
    secs = 0.01
    raw = synth.make_vdif(secs, bw_mhz=bw, nchan=c)
    cfg = o.Config(bw_mhz=bw, nchan=c, freq_res=64, total_s=secs, tscrunch=t)
    f = sigproc.read_fil(o.channelise(raw, cfg))
    assert f.header["tsamp"] == pytest.approx(c * t / bw * 1e-6)
    assert f.header["foff"] == pytest.approx(-bw / c)
    nblocks = int(secs * 2e6 * bw) // (2 * c * 64)
    assert f.data.shape == (nblocks * 64 // t, 1, c)


def test_coherency_products():
    """pol1 = i * pol0 (analytic-signal sense) => PP = QQ, Re PQ* = 0, Im PQ* = -/+ PP."""
    c, r = 8, 32
    n = 2 * c * r
    rng = np.random.default_rng(3)
    spec = np.zeros(n // 2 + 1, complex)
    spec[1:n // 2] = rng.standard_normal(n // 2 - 1) + 1j * rng.standard_normal(n // 2 - 1)
    x0 = np.fft.irfft(spec, n)
    x1 = np.fft.irfft(1j * spec, n)
    prod = o.detect(o.filterbank_block(np.vstack([x0, x1]), c, r), 4)
    pp, qq, re, im = prod
    np.testing.assert_allclose(pp, qq, rtol=1e-9, atol=1e-9 * pp.max())
    assert np.abs(re).max() < 1e-9 * pp.max()
    np.testing.assert_allclose(im, -pp, rtol=1e-9, atol=1e-9 * pp.max())


def test_rescale_and_digitise_rules():
    p = np.array([[[1.0, 2.0, 3.0, 4.0], [5.0, 5.0, 5.0, 5.0]]])
    off, sc = o.rescale_stats(p)
    assert off[0, 0] == -2.5 and sc[0, 0] == pytest.approx(1 / np.sqrt(1.25))
    assert off[0, 1] == -5.0 and sc[0, 1] == 1.0          # zero variance -> scale 1
    x = np.array([-10.0, -6.0, 0.0, 0.0235, 6.0, 10.0])
    assert list(o.digitise_values(x, 8)) == [0, 0, 128, 128, 255, 255]
    assert list(o.digitise_values(np.array([-2.1, -0.6, 0.4, 1.6]), 2)) == [0, 1, 2, 3]
    assert o.digitise_values(np.array([1e30]), 16)[0] == 65535


def test_vdif_epoch_and_tstart():
    assert o.vdif_epoch_mjd(0) == 51544 and o.vdif_epoch_mjd(1) == 51726 and o.vdif_epoch_mjd(40) == 58849
    raw = synth.make_vdif(0.01, bw_mhz=16.0, nchan=32, seconds0=86400 * 3 + 43200)
    cfg = o.Config(bw_mhz=16.0, nchan=32, freq_res=64, total_s=0.004, start_s=0.004)
    f = sigproc.read_fil(o.channelise(raw, cfg))
    assert f.header["tstart"] == pytest.approx(58849 + 3.5 + 0.004 / 86400.0, abs=1e-10)


# ---- known-answer test 7: coherent dedispersion -------------------------------------
@pytest.mark.parametrize("bw", [16.0, -16.0])
def test_dispersed_bursts_realign_at_the_channel_centre(bw):
    """Tone bursts emitted together at three sky frequencies of one channel arrive with the cold-plasma
    delay DM/(2.41e-4 nu^2) (built in the TIME domain, independent of the kernel's conventions); after
    the -F C:D filterbank all three peak where the channel-centre burst does, in either sideband."""
    c, freq, dm, k = 16, 316.0, 1.0, 5
    r, pos, neg, keep = o.coherent_geometry(freq, bw, c, 0, 1, dm)
    assert (r, keep) == (2048, r - pos - neg) and pos > 100 and neg > 100
    n, rate = 2 * c * r, 2.0e6 * abs(bw)
    nu0 = o.channel_centres_sky(freq, bw, c)[k]
    kern = o.chirp(freq, bw, c, r, dm)
    t = np.arange(n) / rate
    raw, ded = [], []
    for dnu in (-0.3, 0.0, 0.3):
        nu = nu0 + dnu
        tau = dm / o.DM_DISPERSION * (1.0 / nu ** 2 - 1.0 / nu0 ** 2)
        fb = (nu - (freq - abs(bw) / 2)) * 1e6 if bw > 0 else ((freq + abs(bw) / 2) - nu) * 1e6
        x = np.exp(-0.5 * ((t - 1.0e-3 - tau) / 5e-6) ** 2) * np.cos(2 * np.pi * fb * (t - tau) + 0.3)
        xx = np.stack([x, np.zeros_like(x)])
        for y, dst in ((o.filterbank_block(xx, c, r)[0, k], raw),
                       (o.filterbank_block_coherent(xx, c, r, kern)[0, k], ded)):
            pw = np.abs(y) ** 2
            dst.append((pw * np.arange(r)).sum() / pw.sum())
    assert raw[0] - raw[1] > 70 and raw[1] - raw[2] > 70          # ~ +-80 us at 1 us per sample: lower arrives later
    assert np.ptp(ded) < 0.05 and abs(ded[1] - raw[1]) < 0.05     # all on the centre's arrival time


def test_coherent_kernel_is_hermitian_safe_and_unit_modulus():
    h = o.chirp(1400.0, 32.0, 64, 256, 56.7)
    assert np.allclose(np.abs(h), 1.0) and h[0, 0] == 1.0
    # zero DM: the coherent path is the plain filterbank with the block edges cut
    x = np.random.default_rng(1).standard_normal((2, 2 * 8 * 64))
    a = o.filterbank_block(x, 8, 64)
    b = o.filterbank_block_coherent(x, 8, 64, o.chirp(1400.0, 32.0, 8, 64, 0.0))
    np.testing.assert_allclose(a, b, atol=1e-9)


def test_overlap_save_blocks_are_seamless():
    """The stream cut into short overlapping blocks (R = 2048) and into long ones (R = 8192) gives the same
    channel time series sample for sample (correlation ~1; a one-sample misplacement of any block would
    decorrelate white noise completely).  Not exactly equal: the brick-wall channels ring at block edges."""
    raw = synth.make_vdif(0.02, bw_mhz=16.0, nchan=16)
    small = o.Config(bw_mhz=16.0, nchan=16, freq_mhz=316.0, dm=1.0, coherent=True, total_s=0.02, freq_res=2048)
    big = o.Config(bw_mhz=16.0, nchan=16, freq_mhz=316.0, dm=1.0, coherent=True, total_s=0.02, freq_res=8192)
    ps, _, s0s, _ = o.detected_power(raw, small)
    pb, _, s0b, _ = o.detected_power(raw, big)
    assert s0s == s0b and ps.shape[2] > pb.shape[2] > 0            # same first sample, more (shorter) blocks
    n = pb.shape[2]
    a, b = ps[0, :, :n] / 2048.0 ** 2, pb[0] / 8192.0 ** 2          # unnormalised transforms: power ~ R^2
    for k in range(16):
        assert np.corrcoef(a[k], b[k])[0, 1] > 0.995
        assert abs(np.corrcoef(a[k, 1:], b[k, :-1])[0, 1]) < 0.1
    assert np.median(np.abs(a - b)) < 0.03 * b.mean()
