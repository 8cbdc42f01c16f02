"""GPU tests of the stages behind the filterbank: HIP dedispersion / fold kernels against oracle/post_oracle.py
(bit-exact on integer rows), at BASELINE size (10 s x 1024 channels) through size-independent properties, and the
known-pulsar flow end to end (config 5: dispersed pulse train -> coherent filterbank -> .fil -> fold -> one phase bin)."""
import contextlib
import io
import json
import os
import time

import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import post, process_vdif as pv, sigproc, synth, vdif
from oracle import post_oracle as po
from tests.test_post import DM0, HDR, P0, as_fil, pulse_train_rows

pytestmark = pytest.mark.gpu

HDR1K = dict(HDR, nchans=1024, foff=-0.03125, fch1=1416.0 - 0.015625, tsamp=32e-6)


@pytest.mark.parametrize("nbits,zerodm,clip", [(8, True, 5.0), (8, False, 0.0), (16, True, 4.0), (32, True, 5.0)])
def test_dedisperse_matches_oracle(hip_lib, nbits, zerodm, clip):
    x = pulse_train_rows(20000, HDR1K).astype(np.float64)
    x[7000:7006] += 80
    data = x.astype(np.uint8) if nbits == 8 else ((x * 55).astype(np.uint16) if nbits == 16 else (x * 0.37 - 3.0).astype(np.float32))
    dms = [0.0, 26.7, DM0, 348.8]
    got, nclip = post.dedisperse(as_fil(data, HDR1K, nbits), dms, zerodm=zerodm, clip=clip, lib=hip_lib)
    want, nclip2 = po.dedisperse(data, fch1=HDR1K["fch1"], foff=HDR1K["foff"], tsamp=HDR1K["tsamp"], dms=dms, zerodm=zerodm,
                                 clip=clip, integer=nbits != 32)
    assert nclip == nclip2 and (clip == 0 or nclip >= 6)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nbits,apply_delays", [(8, False), (8, True), (16, True), (32, True)])
def test_fold_matches_oracle(hip_lib, nbits, apply_delays):
    x = pulse_train_rows(40000, HDR1K)
    data = x if nbits == 8 else ((x.astype(np.uint16) * 201) if nbits == 16 else (x * 0.5).astype(np.float32))
    par = dict(F0=1.0 / P0, F1=-2.5e-9, PEPOCH=HDR1K["tstart"] - 300.0, DM=DM0, PSR="J0000+00")
    prof, hits, nbin = post.fold(as_fil(data, HDR1K, nbits), par, nbin=256, subint_s=0.5, apply_delays=apply_delays, lib=hip_lib)
    wp, wh = po.fold(data, fch1=HDR1K["fch1"], foff=HDR1K["foff"], tsamp=HDR1K["tsamp"], tstart_mjd=HDR1K["tstart"], f0=par["F0"],
                     f1=par["F1"], pepoch_mjd=par["PEPOCH"], dm=DM0, nbin=256, subint_s=0.5, apply_delays=apply_delays)
    assert np.array_equal(hits, wh)
    if nbits == 32:
        np.testing.assert_allclose(prof, wp, rtol=1e-12)
    else:
        assert np.array_equal(prof, wp)


def test_full_size_scan_rows(hip_lib):
    """10 s of a 32 MHz IF as the channeliser writes it (312 500 rows x 1024 channels, 8 bit): totals that any correct
    dedispersion / fold must conserve, plus the wall-clock of both stages (printed; profiles/r02_post_stages.json)."""
    rng = np.random.default_rng(7)
    nrows, nchan = 312500, 1024
    data = rng.integers(100, 156, size=(nrows, nchan), dtype=np.uint8)
    fil = as_fil(data, HDR1K, 8)
    t0 = time.perf_counter()
    y, nclip = post.dedisperse(fil, [0.0, DM0], zerodm=False, clip=0.0, lib=hip_lib)
    t_dd = time.perf_counter() - t0
    # DM 0: the series is the row sum; DM 56.7: the same samples re-aligned: totals equal up to the edge that is cut
    assert np.array_equal(y[0], data[: y.shape[1]].sum(axis=1, dtype=np.int64).astype(np.float32))
    d = po.delays_samples(HDR1K["fch1"], HDR1K["foff"], nchan, HDR1K["tsamp"], DM0)
    want_total = sum(int(data[d[c]: d[c] + y.shape[1], c].sum(dtype=np.int64)) for c in range(nchan))
    assert int(y[1].astype(np.float64).sum()) == want_total
    yz, _ = post.dedisperse(fil, [DM0], zerodm=True, clip=0.0, lib=hip_lib)
    assert abs(float(yz[0].astype(np.float64).mean())) < 0.5                   # zero-DM: the mean is gone
    par = dict(F0=1.0 / P0, F1=0.0, PEPOCH=None, DM=DM0, PSR="x")
    t0 = time.perf_counter()
    prof, hits, nbin = post.fold(fil, par, nbin=512, subint_s=10.0, apply_delays=True, lib=hip_lib)
    t_fold = time.perf_counter() - t0
    assert prof.shape == (1, nchan, 512) and int(hits.sum()) == nrows * nchan
    assert np.array_equal(prof.sum(axis=2)[0], data.sum(axis=0, dtype=np.int64).astype(np.float64))   # per-channel totals
    stats = {"rows": nrows, "nchan": nchan, "dedisperse_2dm_s_host_inclusive": round(t_dd, 4), "fold_s_host_inclusive": round(t_fold, 4)}
    print("POST-STAGES " + json.dumps(stats))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(stats, open("gpurun_out/post_stages.json", "w"))


def _dispersed_pulsar_vdif(secs, bw, freq, dm, period, seed=11):
    """2-bit VDIF of noise amplitude-modulated by a pulse train and dispersed by the ISM transfer function over the USB
    band [freq - bw/2, freq + bw/2] (group delay DM / 2.41e-4 (nu^-2 - nu_top^-2))"""
    rate = 2.0e6 * bw
    n = int(round(secs * rate))
    rng = np.random.default_rng(seed)
    t = np.arange(n) / rate
    ph = (t / period) % 1.0
    env = ((ph > 0.40) & (ph < 0.40 + 150e-6 / period)).astype(np.float64)
    nu = (freq - bw / 2.0) + np.fft.rfftfreq(n, d=1.0 / rate) / 1.0e6          # sky frequency of the rfft bins, MHz
    nu_top = freq + bw / 2.0
    phi = 2.0 * np.pi * 1.0e6 * (dm / 2.41e-4) * (1.0 / nu + nu / nu_top ** 2)
    h = np.exp(1j * (phi - phi[-1]))
    st = np.empty((2, n), np.uint8)
    for pol in range(2):
        x = rng.standard_normal(n) * (1.0 + 3.0 * env)
        x = np.fft.irfft(np.fft.rfft(x) * h, n)
        st[pol] = synth.quantise_2bit(x / x.std())
    spf = 16000
    nfr = n // spf
    return vdif.frame_payload(vdif.pack_states(st[:, : nfr * spf]), bw_mhz=bw, seconds0=1000, ref_epoch=40)


def test_known_pulsar_flow_end_to_end(hip_lib, tmp_path):
    """BASELINE config 5 at small scale: `-D 56.7 -F128:D` coherent filterbank of a dispersed pulse train, then the fold
    of the .fil with a .par file (base2fil.sh:465-493): the pulse lands in ONE phase bin of the dedispersed profile."""
    bw, freq, nchan = 16.0, 1400.0, 128
    raw = _dispersed_pulsar_vdif(1.0, bw, freq, DM0, P0)
    vd = str(tmp_path / "pr001a_ef_no0001_IF1.vdif")
    raw.tofile(vd)
    hdr = pv.make_hdr("J0000+00", freq, vd, pol=2, usb=True, ra="00:00:00", dec="00:00:00", bw=bw, telescope="effelsberg")
    with contextlib.redirect_stdout(io.StringIO()):
        fil = pv.run_digifil(hdr, str(tmp_path), 0, 1.0, nchan, overwrite=True, pol=2, nbit=8, dm=DM0, coherent=True)
    par = tmp_path / "J0000+00.psrcat.par"
    par.write_text("PSRJ J0000+00\nP0 %.6f\nDM %.1f\n" % (P0, DM0))
    ar, profile = post.fold_fil(fil, str(par), nbin=128, subint_s=10.0, lib=hip_lib)
    peak = int(np.argmax(profile))
    off = np.delete(profile, [(peak - 1) % 128, peak, (peak + 1) % 128])
    snr = (profile[peak] - off.mean()) / off.std()
    print("known-pulsar flow: peak bin", peak, "S/N", round(float(snr), 1))
    assert snr > 25
    # one bin: the neighbours hold less than a third of the peak's excess (150 us pulse, 261 us bins, edge-on at worst)
    assert max(profile[(peak - 1) % 128], profile[(peak + 1) % 128]) - off.mean() < 0.6 * (profile[peak] - off.mean())
    assert profile[(peak + 2) % 128] - off.mean() < 0.1 * (profile[peak] - off.mean())
    # the same through the GPU prepdata stage: the series at DM 56.7 peaks once per period
    out = post.prepdata_gpu(fil, DM0, zerodm=False, clip=0, lib=hip_lib)
    y = np.fromfile(out[0], dtype="<f4")
    tsamp = sigproc.read_fil(fil).header["tsamp"]
    hi = np.nonzero(y > np.median(y) + 0.5 * (y.max() - np.median(y)))[0]
    starts = hi[np.insert(np.diff(hi) > 100, 0, True)]
    assert len(starts) >= 25 and np.all(np.abs(np.diff(starts) * tsamp - P0) < 3 * tsamp)


def test_config5_as_stated(hip_lib, tmp_path):
    """BASELINE.json configs[4] at its own shape: 32 MHz at 1.4 GHz, `-F2048:4096 -D 56.7 -F2048:D` (the golden command
    line of the harness), 2 coherent blocks with overlap-save, then the fold of the .fil with a .par file."""
    bw, freq, nchan = 32.0, 1400.0, 2048
    raw = _dispersed_pulsar_vdif(0.7, bw, freq, DM0, P0, seed=21)
    vd = str(tmp_path / "pr001a_ef_no0001_IF1.vdif")
    raw.tofile(vd)
    hdr = pv.make_hdr("J0000+00", freq, vd, pol=2, usb=True, ra="00:00:00", dec="00:00:00", bw=bw, telescope="effelsberg")
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        fil = pv.run_digifil(hdr, str(tmp_path), 0, 0.7, nchan, overwrite=True, pol=2, nbit=8, dm=DM0, coherent=True)
    assert "-d1 -F2048:4096 -D 56.7 -F2048:D" in log.getvalue()
    f = sigproc.read_fil(fil)
    assert f.header["nchans"] == 2048 and f.header["refdm"] == DM0 and f.data.shape[0] >= 2 * 4000
    par = tmp_path / "J0000+00.psrcat.par"
    par.write_text("PSRJ J0000+00\nP0 %.6f\nDM %.1f\n" % (P0, DM0))
    ar, profile = post.fold_fil(fil, str(par), nbin=128, subint_s=10.0, lib=hip_lib)
    peak = int(np.argmax(profile))
    off = np.delete(profile, [(peak - 1) % 128, peak, (peak + 1) % 128])
    snr = (profile[peak] - off.mean()) / off.std()
    print("config 5 as stated: peak bin", peak, "S/N", round(float(snr), 1))
    assert snr > 20
    assert profile[(peak + 2) % 128] - off.mean() < 0.15 * (profile[peak] - off.mean())
    assert profile[(peak - 2) % 128] - off.mean() < 0.15 * (profile[peak] - off.mean())


@pytest.mark.parametrize("nbits,zerodm,clip", [(8, True, 5.0), (16, False, 0.0), (32, True, 4.0)])
def test_dm_range_on_the_tiled_kernel_matches_oracle(hip_lib, nbits, zerodm, clip):
    """prepsubband's DM range (process_vdif.py:202-229: -lodm / -numdms / -dmstep): 64 DMs from the same rows through the
    LDS-tiled kernel (DM groups of 8, 64-byte channel tiles) -- bit-identical to the oracle, float rows included (double
    sums in ascending channel order), with clipped samples and the zero-DM term staged in the LDS as well"""
    x = pulse_train_rows(12000, HDR1K).astype(np.float64)
    x[5000:5005] += 90
    data = x.astype(np.uint8) if nbits == 8 else ((x * 55).astype(np.uint16) if nbits == 16 else (x * 0.37 - 3.0).astype(np.float32))
    dms = post.dm_list(20.0, 83.0, 1.0)
    assert len(dms) == 64
    got, nclip = post.dedisperse(as_fil(data, HDR1K, nbits), dms, zerodm=zerodm, clip=clip, lib=hip_lib)
    want, nclip2 = po.dedisperse(data, fch1=HDR1K["fch1"], foff=HDR1K["foff"], tsamp=HDR1K["tsamp"], dms=dms, zerodm=zerodm,
                                 clip=clip, integer=nbits != 32)
    assert nclip == nclip2
    assert np.array_equal(got, want)


def test_dm_range_device_time_at_full_size(hip_lib):
    """320 MB of rows (10 s x 1024 channels, 8 bit) x 64 DMs, rows resident in HBM: device time of frbch_dedisperse_device
    (< 60 ms) and conservation of the total (every output sample is a sum of nchan input samples)"""
    import ctypes as C
    from frb_baseband_amd import _lib
    from tests.hipmem import DeviceBuffer
    rng = np.random.default_rng(11)
    nrows, nchan = 312500, 1024
    data = rng.integers(100, 156, size=(nrows, nchan), dtype=np.uint8)
    fil = as_fil(data, HDR1K, 8)
    dms = np.asarray(post.dm_list(300.0, 363.0, 1.0), dtype=np.float64)
    desc = post.fil_desc(fil.header)
    nout = hip_lib.frbch_dedisperse_nout(C.byref(desc), nrows, dms.ctypes.data, len(dms))
    assert nout > 0
    d_rows = DeviceBuffer.from_numpy(data)
    d_out = DeviceBuffer(len(dms) * nout * 4)
    err = C.create_string_buffer(256)
    nclip = C.c_uint64(0)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc = hip_lib.frbch_dedisperse_device(C.byref(desc), d_rows.ptr, nrows, dms.ctypes.data, len(dms), 0, 0.0, 0, d_out.ptr, nout,
                                             C.byref(nclip), err, len(err))
        times.append(time.perf_counter() - t0)
        assert rc == 0, err.value
    y = d_out.to_numpy(np.float32).reshape(len(dms), nout)
    # every series: sum over channels of shifted rows -- its mean is nchan x the mean sample (to the noise of the shifts)
    assert abs(y.mean() / nchan - data.mean()) < 0.05
    # spot check against the definition for a few (dm, t)
    from oracle import post_oracle
    for i in (0, 31, 63):
        dly = post_oracle.delays_samples(HDR1K["fch1"], HDR1K["foff"], nchan, HDR1K["tsamp"], float(dms[i]))
        for t in (0, 1234, nout - 1):
            assert y[i, t] == float(data[t + dly, np.arange(nchan)].astype(np.int64).sum())
    stats = {"rows": nrows, "nchan": nchan, "ndm": len(dms), "call_s_best": min(times), "call_s_all": times}
    print("DEDISP-RANGE " + json.dumps(stats))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/post_dm_range.json", "w") as f:
        json.dump(stats, f)
    assert min(times) < 0.060, stats
