"""Shared parity checks: HIP path (or the test-only emulator) against the fp64 oracle."""
import numpy as np

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import sigproc, synth
from oracle import frb_oracle as o

# Tolerances (stated, see DESIGN.md "Parity"):
#  * detected power: |P - P_oracle| <= POWER_RTOL * mean power of that (product, channel) series
#    (fp32 FFT of up to 2^26 points against an fp64 oracle).
#  * digitised codes: identical except where the oracle's pre-rounding value lies within
#    TIE_EPS_SIGMA (in units of the rescaled sigma, i.e. TIE_EPS_SIGMA * digi_scale code units:
#    2e-3 of an 8-bit code, 0.5 of a 16-bit code) of a rounding boundary; such samples may differ
#    by 1 and must stay below MISMATCH_FRAC_PER_SIGMA * digi_scale of all samples (2e-4 for 8 bit).
POWER_RTOL = 2e-5
TIE_EPS_SIGMA = 1.0e-4
MISMATCH_FRAC_PER_SIGMA = 1.0e-5
CODE_TIE_EPS = TIE_EPS_SIGMA * 127.5 / 6.0          # 8-bit values, kept for reference
CODE_MISMATCH_FRAC = MISMATCH_FRAC_PER_SIGMA * 127.5 / 6.0
RESCALE_RTOL = 2e-6


def make_case(bw, nchan, secs, **kw):
    raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan, **{k: kw.pop(k) for k in list(kw) if k in ("payload_bytes", "legacy", "if_index")})
    return raw


def oracle_cfg(bw, nchan, secs, pol=2, nbit=8, tscr=1, interval=10.0, const=1, freq_res=0, start=0.0,
               dm=0.0, coherent=0, freq=1608.0, levels=None):
    return o.Config(bw_mhz=bw, nchan=nchan, total_s=secs, start_s=start, pol_mode=pol, nbit=nbit,
                    tscrunch=tscr, rescale_interval_s=interval, rescale_constant=bool(const),
                    freq_res=freq_res, source="unknown", telescope="ONSALA85", dm=dm, coherent=bool(coherent),
                    freq_mhz=freq, levels=levels)


def lib_cfg(lib, bw, nchan, secs, pol=2, nbit=8, tscr=1, interval=10.0, const=1, freq_res=0, start=0.0, maxb=0, flags=0,
            dm=0.0, coherent=0, freq=1608.0, levels=None):
    kw = {} if levels is None else {"levels": levels}
    return ch.new_config(lib, bw_mhz=bw, nchan=nchan, total_s=secs, start_s=start, pol_mode=pol,
                         nbit_out=nbit, tscrunch=tscr, rescale_interval_s=interval,
                         rescale_constant=const, freq_res=freq_res, max_blocks_per_launch=maxb, flags=flags,
                         dm=dm, coherent=coherent, freq_mhz=freq, **kw)


def expected_boundary_distance(ocfg):
    """distance of the oracle's pre-rounding value t+0.5 from the nearest integer, per output sample"""
    mean, scale, vmax = o.digi_params(ocfg.nbit)
    x = ocfg.result["rescaled"]
    if ocfg.bw_mhz > 0:
        x = x[:, ::-1, :]
    t = x.transpose(2, 0, 1) * scale + mean + 0.5
    return np.abs(t - np.round(t))


def check_codes(ref_bytes, got_bytes, ocfg):
    fr = sigproc.read_fil(ref_bytes)
    fg = sigproc.read_fil(got_bytes)
    assert ref_bytes[:fr.header_bytes] == got_bytes[:fg.header_bytes], "SIGPROC header differs"
    assert fr.data.shape == fg.data.shape
    if ocfg.nbit == -32:
        ref = fr.data.astype(np.float64)
        np.testing.assert_allclose(fg.data, ref, rtol=0, atol=POWER_RTOL * 50 * max(1.0, np.abs(ref).max()))
        return 0
    d = fg.data.astype(np.int64) - fr.data.astype(np.int64)
    bad = np.nonzero(d)
    nbad = bad[0].size
    if nbad:
        assert np.abs(d).max() <= 1, "code differs by more than 1"
        _mean, dscale, _vmax = o.digi_params(ocfg.nbit)
        dist = expected_boundary_distance(ocfg)[bad]
        assert dist.max() <= TIE_EPS_SIGMA * dscale, f"mismatch away from a rounding tie: {dist.max()}"
        assert nbad <= max(2, MISMATCH_FRAC_PER_SIGMA * dscale * d.size), f"{nbad} of {d.size} codes differ"
    return nbad


def run_streaming_case(lib, bw, nchan, secs, **kw):
    """oracle .fil vs library .fil through push/flush/pull; returns mismatch count"""
    kw = dict(kw)
    gen = {k: kw.pop(k) for k in ("bits", "payload_bytes", "legacy") if k in kw}    # input-format variants
    raw = synth.make_vdif(secs + kw.get("start", 0.0), bw_mhz=abs(bw), nchan=nchan, **gen)
    ocfg = oracle_cfg(bw, nchan, secs, **{k: v for k, v in kw.items() if k not in ("maxb", "flags")})
    ref = o.channelise(raw, ocfg)
    cfg = lib_cfg(lib, bw, nchan, secs, **kw)
    with ch.Channeliser(cfg, lib) as c:
        got = c.channelise_bytes(raw)
        resc = c.get_rescale() if c.get_info().have_rescale else None
    nbad = check_codes(ref, got, ocfg)
    if resc is not None and kw.get("interval", 10.0) > 0 and kw.get("const", 1) and "offset0" in ocfg.result:
        off0, sc0 = ocfg.result["offset0"], ocfg.result["scale0"]
        assert np.abs(resc[0] - off0).max() <= RESCALE_RTOL * np.abs(off0).max()
        np.testing.assert_allclose(resc[1], sc0, rtol=RESCALE_RTOL)
    return nbad
