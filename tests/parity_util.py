"""Shared parity checks: HIP path (or the test-only emulator) against the fp64 oracle."""
import numpy as np

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import sigproc, synth
from oracle import frb_oracle as o

# Tolerances (stated, see DESIGN.md "Parity"):
#  * detected power: |P - P_oracle| <= power_rtol(...) * mean TOTAL power (PP+QQ) of that channel's series (fp32 FFTs of
#    2^17 .. 2^26 points against an fp64 oracle): a regression guard from the error model of an fp32 FFT chain; the parity
#    statement is the relation to the fp32 CPU port (tests/test_gpu_pins.py::test_power_error_distribution).
#  * digitised codes: identical except where the oracle's pre-rounding value lies within
#    TIE_EPS_SIGMA (in units of the rescaled sigma, i.e. TIE_EPS_SIGMA * digi_scale code units:
#    2e-3 of an 8-bit code, 0.5 of a 16-bit code) of a rounding boundary; such samples may differ
#    by 1 and must stay below MISMATCH_FRAC_PER_SIGMA * digi_scale of all samples (2e-4 for 8 bit).
POWER_RTOL = 1.1e-5     # the loosest bound any configuration needs (kept for callers that do not know N)

# Per-configuration REGRESSION GUARD (not a parity statement): an fp32 FFT chain of N = 2 C R points against the fp64 oracle
# errs like eps * sqrt(log2 N), and a sum over T scrunched samples averages the (independent) errors down by sqrt(T); the model
# 1.02e-6 sqrt(log2 N) / sqrt(min(T, 8)) reproduces the maxima measured on MI355X (profiles/r02_power_error_distribution.jsonl)
# to +-25 %, the bound is twice it.  What the error MEANS is stated by tests/test_gpu_pins.py::test_power_error_distribution:
# the HIP chain's 99.9 % point and maximum stay within 1.5 x those of the fp32 CPU port (oracle/frb_oracle.c) of the same
# transform -- the closest thing here to the reference's fp32 CPU program (DSPSR itself is absent: parity unpinned).
def power_rtol(nchan, freq_res, tscr=1):
    import math
    log2n = int(round(math.log2(2 * nchan * freq_res)))
    t = min(int(tscr), 8)
    return min(POWER_RTOL, 2.0 * 1.02e-6 * math.sqrt(log2n) / math.sqrt(t))


TIE_EPS_SIGMA = 1.0e-4
MISMATCH_FRAC_PER_SIGMA = 1.0e-5
CODE_TIE_EPS = TIE_EPS_SIGMA * 127.5 / 6.0          # 8-bit values, kept for reference
CODE_MISMATCH_FRAC = MISMATCH_FRAC_PER_SIGMA * 127.5 / 6.0
RESCALE_RTOL = 2e-6


def make_case(bw, nchan, secs, **kw):
    raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan, **{k: kw.pop(k) for k in list(kw) if k in ("payload_bytes", "legacy", "if_index")})
    return raw


def oracle_cfg(bw, nchan, secs, pol=2, nbit=8, tscr=1, interval=10.0, const=1, freq_res=0, start=0.0,
               dm=0.0, coherent=0, freq=1608.0, levels=None, dynamic=None):
    if dynamic is not None:   # dynamic level setting: dict of nsample / cutoff_sigma / threshold (missing = the defaults)
        dynamic = dict(nsample=dynamic.get("nsample", o.DLS_NSAMPLE), cutoff_sigma=dynamic.get("cutoff_sigma", o.DLS_CUTOFF_SIGMA),
                       threshold=dynamic.get("threshold", o.DLS_THRESHOLD))
    return o.Config(bw_mhz=bw, nchan=nchan, total_s=secs, start_s=start, pol_mode=pol, nbit=nbit,
                    tscrunch=tscr, rescale_interval_s=interval, rescale_constant=bool(const),
                    freq_res=freq_res, source="unknown", telescope="ONSALA85", dm=dm, coherent=bool(coherent),
                    freq_mhz=freq, levels=levels, dynamic=dynamic)


def lib_cfg(lib, bw, nchan, secs, pol=2, nbit=8, tscr=1, interval=10.0, const=1, freq_res=0, start=0.0, maxb=0, flags=0,
            dm=0.0, coherent=0, freq=1608.0, levels=None, dynamic=None):
    kw = {} if levels is None else {"levels": levels}
    if dynamic is not None:
        kw.update(unpack_mode=1, dls_nsample=dynamic.get("nsample", 0), dls_cutoff_sigma=dynamic.get("cutoff_sigma", 0.0),
                  dls_threshold=dynamic.get("threshold", 0.0))
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


def check_code_arrays(ref_data, got_data, ocfg):
    """digitised rows [t][product][chan] of ONE IF against the oracle's (`ocfg` = the oracle run that produced ref_data):
    identical except at rounding ties, as check_codes; returns the number of differing codes"""
    assert ocfg.nbit != -32 and ref_data.shape == got_data.shape
    d = got_data.astype(np.int64) - ref_data.astype(np.int64)
    bad = np.nonzero(d)
    nbad = bad[0].size
    if nbad:
        assert np.abs(d).max() <= 1, "code differs by more than 1"
        _mean, dscale, _vmax = o.digi_params(ocfg.nbit)
        dist = expected_boundary_distance(ocfg)[: ref_data.shape[0]][bad]
        assert dist.max() <= TIE_EPS_SIGMA * dscale, f"mismatch away from a rounding tie: {dist.max()}"
        assert nbad <= max(2, MISMATCH_FRAC_PER_SIGMA * dscale * d.size), f"{nbad} of {d.size} codes differ"
    return nbad


def check_codes(ref_bytes, got_bytes, ocfg):
    fr = sigproc.read_fil(ref_bytes)
    fg = sigproc.read_fil(got_bytes)
    assert ref_bytes[:fr.header_bytes] == got_bytes[:fg.header_bytes], "SIGPROC header differs"
    assert fr.data.shape == fg.data.shape
    if ocfg.nbit == -32:
        # float output x = (P + offset) * scale: the power bound times the (product, channel) scale actually applied
        # (slope of x against P over the series; 1.5x slack where the pair changes per interval), plus fp32 rounding of x
        ref = fr.data.astype(np.float64)
        p = ocfg.result["power"]                                            # [nif][C][nt]
        x = ocfg.result["rescaled"]
        tot = p[:1] if p.shape[0] == 1 or ocfg.pol_mode == 5 else p[0:1] + p[1:2]
        chan_mean = np.abs(tot).mean(axis=2)                                # [1][C]
        sp = p.std(axis=2)
        slope = np.where(sp > 0, x.std(axis=2) / np.where(sp > 0, sp, 1.0), 1.0)   # [nif][C]
        slack = 1.0 if (ocfg.rescale_constant or ocfg.rescale_interval_s <= 0) else 1.5
        rtol = power_rtol(ocfg.nchan, ocfg.result["geometry"][0], ocfg.tscrunch)
        tol = slack * rtol * chan_mean * slope                              # [nif][C]
        if ocfg.bw_mhz > 0:
            tol = tol[:, ::-1]
        err = np.abs(fg.data.astype(np.float64) - ref)
        bound = tol[None, :, :] + 2.5e-7 * np.abs(ref)
        assert np.all(err <= bound), f"float output off by {(err / bound).max():.2f} x the stated bound"
        return 0
    return check_code_arrays(fr.data, fg.data, ocfg)


_ORACLE_CACHE: dict = {}


def run_streaming_case(lib, bw, nchan, secs, **kw):
    """oracle .fil vs library .fil through push/flush/pull; returns mismatch count"""
    kw = dict(kw)
    gen = {k: kw.pop(k) for k in ("bits", "payload_bytes", "legacy") if k in kw}    # input-format variants
    raw = synth.make_vdif(secs + kw.get("start", 0.0), bw_mhz=abs(bw), nchan=nchan, **gen)
    okw = {k: v for k, v in kw.items() if k not in ("maxb", "flags")}
    # cases that differ only in kernel selection (flags) or batching (maxb) share ONE oracle run (the fp64 numpy chain at 2^24 .. 2^26
    # points is most of the GPU suite's wall time); the last few results are kept
    key = (bw, nchan, secs, tuple(sorted((k, (tuple(v) if isinstance(v, (list, tuple)) else (tuple(sorted(v.items())) if isinstance(v, dict) else v))) for k, v in okw.items())), tuple(sorted(gen.items())))
    if key in _ORACLE_CACHE:
        ref, ocfg = _ORACLE_CACHE[key]
    else:
        ocfg = oracle_cfg(bw, nchan, secs, **okw)
        ref = o.channelise(raw, ocfg)
        if len(_ORACLE_CACHE) >= 4:
            _ORACLE_CACHE.pop(next(iter(_ORACLE_CACHE)))
        _ORACLE_CACHE[key] = (ref, ocfg)
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
