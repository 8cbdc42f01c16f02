"""The optional dynamic level setting of the 2-bit unpack (frbch_config.unpack_mode 1).

What it is pinned by: nothing of the reference (DSPSR is absent and the reference passes only the bare `-2`,
process_vdif.py:157,160) -- so the restatement in oracle/frb_oracle.py is checked against the PROPERTIES the scheme is defined by
(closed-form levels, power conservation over a 16-fold range of input power, excision of abnormal windows), and the library
against that restatement: exactly on the unpacked voltages (emulator and GPU), through the whole path like every other case."""
import math

import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import vdif
from oracle import frb_oracle as o
from tests import parity_util as pu


def _digitise(x, thr=o.DLS_THRESHOLD):
    """f64[2][n] voltages -> 2-bit offset-binary payload bytes (two dual-pol samples per byte, as unpack_2bit reads them)"""
    st = np.where(x < -thr, 0, np.where(x < 0, 1, np.where(x < thr, 2, 3))).astype(np.uint8)
    return (st[0, 0::2] | (st[1, 0::2] << 2) | (st[0, 1::2] << 4) | (st[1, 1::2] << 6)).astype(np.uint8)


def _varying_power_payload(nwin=96, nsample=512, seed=3):
    """Gaussian noise whose rms changes from window to window (0.5 .. 2 nominal), a few windows of pure impulse"""
    rng = np.random.default_rng(seed)
    sig = np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=(2, nwin)))
    x = rng.normal(size=(2, nwin, nsample)) * sig[:, :, None]
    x[0, 5] = 50.0          # saturated: every sample high -> k = 0
    x[1, 9] *= 1e-3         # dead: every sample low -> k = nsample
    return _digitise(x.reshape(2, -1)), sig


def test_levels_in_closed_form():
    """the table against an independent evaluation (bisection for erfinv, math.erf): nominal levels and the conservation identity"""
    n = 512
    tab = o.dls_table(n, cutoff_sigma=-1.0).astype(np.float64)
    assert np.all(tab[0] == 0) and np.all(tab[n] == 0) and np.all(tab[1:n] > 0)
    for k in (1, 37, 200, 341, 342, 450, 511):
        phi = k / n
        lo_, hi_ = 0.0, 6.0
        for _ in range(200):          # u with erf(u / sqrt 2) = phi
            mid = 0.5 * (lo_ + hi_)
            if math.erf(mid / math.sqrt(2.0)) < phi:
                lo_ = mid
            else:
                hi_ = mid
        u = 0.5 * (lo_ + hi_)
        s = o.DLS_THRESHOLD / u
        g = math.sqrt(2.0 / math.pi) * u * math.exp(-0.5 * u * u)
        assert tab[k, 0] == pytest.approx(s * math.sqrt(1.0 - g / phi), rel=2e-7)
        assert tab[k, 1] == pytest.approx(s * math.sqrt(1.0 + g / (1.0 - phi)), rel=2e-7)
        # power conservation: Phi lo^2 + (1 - Phi) hi^2 = s^2
        assert phi * tab[k, 0] ** 2 + (1 - phi) * tab[k, 1] ** 2 == pytest.approx(s * s, rel=1e-6)
    # at the nominal threshold (Phi0 = 2/3 of the samples inside) the estimated rms is 1
    k0 = round(n * math.erf(o.DLS_THRESHOLD / math.sqrt(2.0)))
    assert k0 == 341
    assert tab[k0, 0] == pytest.approx(0.5246, abs=2e-3) and tab[k0, 1] == pytest.approx(1.566, abs=3e-3)
    # the default excision keeps counts within 10 standard deviations of the nominal one
    d = o.dls_table(n)
    phi0 = math.erf(o.DLS_THRESHOLD / math.sqrt(2.0))
    sd = math.sqrt(n * phi0 * (1 - phi0))
    kept = np.nonzero(d[:, 0] > 0)[0]
    assert kept[0] == math.ceil(n * math.erf(o.DLS_THRESHOLD / math.sqrt(2.0)) - 10 * sd)
    assert kept[-1] == math.floor(n * math.erf(o.DLS_THRESHOLD / math.sqrt(2.0)) + 10 * sd)


def test_unpacked_power_follows_the_input_power():
    """what the scheme is for: over a 16-fold range of input power the dynamic unpack returns the undigitised power within 1.5 %,
    the static table compresses it (6.1 x .. 1.8 x)"""
    rng = np.random.default_rng(1)
    for sig in (0.5, 0.8, 1.0, 1.3, 2.0):
        b = _digitise(rng.normal(0.0, sig, (2, 1 << 18)))
        dyn = o.unpack_2bit_dynamic(b, cutoff_sigma=-1.0)
        assert (dyn ** 2).mean() / sig ** 2 == pytest.approx(1.0, abs=0.015)
    r = [(o.unpack_2bit(_digitise(rng.normal(0.0, s, (2, 1 << 16)))) ** 2).mean() / s ** 2 for s in (0.5, 2.0)]
    assert r[0] > 3 * r[1]


def test_abnormal_windows_are_zeroed():
    payload, sig = _varying_power_payload()
    v = o.unpack_2bit_dynamic(payload).reshape(2, -1, 512)
    assert np.all(v[0, 5] == 0) and np.all(v[1, 9] == 0)          # saturated / dead windows
    tab = o.dls_table()
    k = (np.abs(o.unpack_2bit(payload)) < 2).reshape(2, -1, 512).sum(axis=2)
    zeroed = np.all(v == 0, axis=2)
    assert np.array_equal(zeroed, tab[k, 0] == 0)
    assert 2 < zeroed.sum() < zeroed.size / 2                      # the power range 0.5 .. 2 crosses the 10-sigma limits


def _tap(lib, payload, dynamic, on_device):
    raw = vdif.frame_payload(payload, bw_mhz=32.0, bits=2)
    nfr = raw.size // 8032
    nsamp = payload.size * 2
    cfg = ch.new_config(lib, bw_mhz=32.0, nchan=64, unpack_mode=1, dls_nsample=dynamic.get("nsample", 0),
                        dls_cutoff_sigma=dynamic.get("cutoff_sigma", 0.0), dls_threshold=dynamic.get("threshold", 0.0))
    with ch.Channeliser(cfg, lib) as c:
        if on_device:
            from tests.hipmem import DeviceBuffer
            d_raw = DeviceBuffer.from_numpy(raw)
            d_v = DeviceBuffer(2 * nsamp * 4)
            c.unpack_device(d_raw.ptr.value, nfr, 8032, 32, 0, nsamp, 0, d_v.ptr.value, d_v.nbytes)
            d_v4 = DeviceBuffer(4 * nsamp * 4)
            with pytest.raises(ch.InputError):     # the register kernels' decoders do not look levels up per window
                c.unpack_device(d_raw.ptr.value, nfr, 8032, 32, 0, nsamp, 1, d_v4.ptr.value, d_v4.nbytes)
            return d_v.to_numpy(np.float32).reshape(2, nsamp)
        volt = np.zeros((2, nsamp), np.float32)
        c.unpack_device(raw.ctypes.data, nfr, 8032, 32, 0, nsamp, 0, volt.ctypes.data, volt.nbytes)
        with pytest.raises(ch.InputError):         # not a whole number of windows
            c.unpack_device(raw.ctypes.data, nfr, 8032, 32, 0, nsamp - 2, 0, volt.ctypes.data, volt.nbytes)
        return volt


DYN = [dict(), dict(nsample=64, cutoff_sigma=3.0), dict(nsample=2048, cutoff_sigma=-1.0, threshold=1.25)]


@pytest.mark.parametrize("dynamic", DYN)
def test_unpack_tap_equals_the_restatement_on_the_emulator(emu_lib, dynamic):
    """window counts through the frame arithmetic (windows straddle the 8000-byte payloads) + the host's level table, bit for bit"""
    payload, _ = _varying_power_payload(nwin=125 * 4, nsample=512)       # 128000 bytes = 16 frames
    want = o.unpack_2bit_dynamic(payload, nsample=dynamic.get("nsample", 512), cutoff_sigma=dynamic.get("cutoff_sigma", 10.0),
                                 threshold=dynamic.get("threshold", o.DLS_THRESHOLD)).astype(np.float32)
    got = _tap(emu_lib, payload, dynamic, False)
    assert np.array_equal(got, want)


def test_invalid_and_missing_frames_with_dynamic_levels(emu_lib):
    """the windows are counted on the payload as it lies in the stream handed to the filterbank (fillers are zero bytes: all samples
    in the outer negative state, so a window inside a filler is zeroed by the excision anyway); samples of flagged frames are zeroed
    AFTER the level lookup, as with the static table (DESIGN.md sections 2a, 3a)"""
    from frb_baseband_amd import synth
    raw = synth.make_vdif(0.012, bw_mhz=16.0, nchan=32).copy()
    marked = raw.copy()
    marked[3 * 8032 + 3] |= 0x80                                          # invalid bit of frame 3
    marked = np.concatenate([marked[: 7 * 8032], marked[9 * 8032:]])     # frames 7 and 8 are missing
    kw = dict(freq_res=64, dynamic=dict(nsample=128, cutoff_sigma=4.0))
    ocfg = pu.oracle_cfg(16.0, 32, 0.012, **kw)
    ref = o.channelise(marked, ocfg)
    assert ocfg.result["frame_counters"] == dict(gaps=1, filled=2, invalid=1)
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.012, **kw), emu_lib) as c:
        got = c.channelise_bytes(marked)
        info = c.get_info()
    assert (info.frames_invalid, info.frame_gaps, info.frames_filled) == (1, 1, 2)
    pu.check_codes(ref, got, ocfg)


def test_configuration_errors(emu_lib):
    for kw in (dict(unpack_mode=2), dict(unpack_mode=1, dls_nsample=100), dict(unpack_mode=1, dls_nsample=8),
               dict(unpack_mode=1, input_bits=1), dict(unpack_mode=1, dls_threshold=-1.0),
               dict(unpack_mode=1, nchan=2, freq_res=2, dls_nsample=16)):      # window longer than a block
        with pytest.raises(ch.InputError):
            ch.Channeliser(ch.new_config(emu_lib, bw_mhz=32.0, **{"nchan": 64, **kw}), emu_lib).close()


def test_shim_spelling(emu_lib):
    """DSPSR spells the unpacker options -2n<nsample> -2c<cutoff> -2t<threshold>; the bare -2 stays the static table"""
    import ctypes as C
    from frb_baseband_amd import _lib
    def parse(extra):
        argv = ["digifil", "-cont", "-c", "-b8", "-S1", "-T10", "-2", "-D", "0.0"] + extra + ["-o", "x.fil", "x.hdr", "-threads", "1", "-d1", "-F128:512"]
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        cfg = _lib.FrbchConfig()
        emu_lib.frbch_config_init(C.byref(cfg))
        hdr, out, err = C.create_string_buffer(256), C.create_string_buffer(256), C.create_string_buffer(256)
        rc = emu_lib.frbch_parse_digifil_argv(len(argv), arr, C.byref(cfg), hdr, 256, out, 256, err, 256)
        assert rc == 0, err.value
        return cfg
    assert parse([]).unpack_mode == 0
    c = parse(["-2n256", "-2c3.5"])
    assert (c.unpack_mode, c.dls_nsample) == (1, 256) and c.dls_cutoff_sigma == pytest.approx(3.5)
    assert parse(["-2t1.1"]).dls_threshold == pytest.approx(1.1)


# ---- GPU --------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dynamic", DYN)
def test_unpack_tap_equals_the_restatement(hip_lib, dynamic):
    payload, _ = _varying_power_payload(nwin=125 * 4, nsample=512)
    want = o.unpack_2bit_dynamic(payload, nsample=dynamic.get("nsample", 512), cutoff_sigma=dynamic.get("cutoff_sigma", 10.0),
                                 threshold=dynamic.get("threshold", o.DLS_THRESHOLD)).astype(np.float32)
    assert np.array_equal(_tap(hip_lib, payload, dynamic, True), want)


@pytest.mark.gpu
@pytest.mark.parametrize("bw,nchan,secs,kw", [
    (32.0, 1024, 0.55, dict(dynamic={})),                                              # configs[1] shape; K2 stays frbch_k2_*
    (-32.0, 1024, 0.3, dict(pol=4, dynamic=dict(nsample=128, cutoff_sigma=2.5))),      # configs[2] products, many windows zeroed
    (16.0, 128, 0.2, dict(nbit=-32, payload_bytes=1000, legacy=1, start=0.05, dynamic=dict(cutoff_sigma=-1.0, threshold=1.2))),
    (16.0, 64, 0.05, dict(dm=5.0, coherent=1, freq=1400.0, dynamic=dict(nsample=64))),
])
def test_fil_matches_oracle(hip_lib, bw, nchan, secs, kw):
    pu.run_streaming_case(hip_lib, bw, nchan, secs, **kw)


@pytest.mark.gpu
def test_the_generic_k1_runs_it_and_says_so(hip_lib):
    """the timing report names the K1 that ran (the per-window levels are looked up by the generic K1 only)"""
    from frb_baseband_amd import synth
    raw = synth.make_vdif(0.3, bw_mhz=32.0, nchan=1024)
    with ch.Channeliser(pu.lib_cfg(hip_lib, 32.0, 1024, 0.3, dynamic={}), hip_lib) as c:
        c.set_profiling(True)
        c.channelise_bytes(raw)
        names = [k for k, v in c.get_timing().items() if v["launches"]]
    assert "frbch_k1_branch" in names and not any(n.startswith("frbch_k1_wave") for n in names), names
