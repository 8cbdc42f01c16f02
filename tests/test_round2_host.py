"""Round-2 host-side checks that need no GPU: ABI v2 fields, flag hygiene, error classes, the IQUV and level-table
extensions and the unpack tap through the TEST-ONLY emulator build, scan buffer sizing of the push-driven fallback."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from frb_baseband_amd import _lib, channeliser as ch, digifil_args, multi_if, process_vdif as pv, sigproc, synth, vdif
from oracle import frb_oracle as o
from tests import parity_util as pu


def test_error_classes_share_the_reference_base():
    """process_vdif.py:227-255: InputError and RunError derive from Error"""
    assert issubclass(pv.InputError, pv.Error) and issubclass(pv.RunError, pv.Error)
    with pytest.raises(pv.Error):
        pv.run_digifil("/d/x.hdr", "/fifo", overwrite=True, nbit=7)
    with pytest.raises(pv.Error):
        raise pv.RunError("x")
    assert pv.Error is ch.Error


def test_iquv_is_an_extension_not_a_change_of_the_reference_surface():
    with pytest.raises(pv.InputError, match="pol = 5 not implemented"):       # golden message, unchanged
        pv.digifil_command("/d/x.hdr", "/o/x.fil", 0, 1, 128, 5, 8, 1, 1, 0.0, False, False)
    cmd = pv.digifil_command("/d/x.hdr", "/o/x.fil", 0, 1, 128, 4, 8, 1, 1, 0.0, False, False, iquv=True)
    assert cmd.endswith("-d4 -F128:512 -iquv")
    cfg, _h, _o = digifil_args.parse(cmd, read_hdr=False)
    assert cfg.pol_mode == 5
    with pytest.raises(ch.InputError, match="-iquv needs -d4"):
        digifil_args.parse(cmd.replace("-d4", "-d1"), read_hdr=False)
    with pytest.raises(pv.InputError):
        pv.digifil_command("/d/x.hdr", "/o/x.fil", 0, 1, 128, 2, 8, 1, 1, 0.0, False, False, iquv=True)


def test_hdr_values_that_do_not_fit_are_argument_errors(tmp_path, hip_lib):
    long_name = str(tmp_path / ("d" * 600 + ".vdif"))
    hdr = tmp_path / "x.hdr"
    hdr.write_text("HDR_VERSION 0.1\nTELESCOPE  Ef\nSOURCE     R3\nRA         01:00:00\nDEC        02:00:00\n"
                   "FREQ       1400.0\nBW         32.0\nDATAFILE   %s\nINSTRUMENT VDIF\nNPOL       2" % long_name)
    cfg = ch.new_config(hip_lib)
    assert hip_lib.frbch_config_from_hdr(str(hdr).encode(), C.byref(cfg)) == _lib.E_ARG


def test_unknown_flag_bits_are_rejected(emu_lib):
    # ablations (8..19), rejected kernel variants / layouts / lane modes kept for A/B runs (4 .. 128, 21, 23 .. 26): experiments builds only
    for bad in (1 << 8, 1 << 12, 1 << 19, 4, 8, 16, 32, 64, 128, 1 << 21, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 31):
        with pytest.raises(ch.InputError, match="unknown bit"):
            ch.Channeliser(ch.new_config(emu_lib, flags=bad), emu_lib)
    for ov in (160 | (1 << 24), 192 | (2 << 24), 176 | (3 << 24) | (2 << 16)):      # CU-masked lanes, forced batching
        cfg = ch.new_config(emu_lib)
        cfg.overlap = ov
        with pytest.raises(ch.InputError, match="overlap"):
            ch.Channeliser(cfg, emu_lib)
    # the four production switches (and, in this test-only build, the whole-file paths without their threads)
    ch.Channeliser(ch.new_config(emu_lib, flags=1 | 2 | (1 << 20) | (1 << 27) | (1 << 22)), emu_lib).close()
    ch.Channeliser(ch.new_config(emu_lib, flags=1 << 28), emu_lib).close()
    with pytest.raises(ch.InputError, match="buffered AND"):
        ch.Channeliser(ch.new_config(emu_lib, flags=(1 << 27) | (1 << 28)), emu_lib)


def test_product_library_is_not_an_experiments_build(hip_lib):
    assert b"experiments" not in hip_lib.frbch_version()
    assert b"abi 5" in hip_lib.frbch_version()


@pytest.mark.parametrize("bw,nchan,secs,kw", [
    (16.0, 32, 0.03, dict(pol=5, freq_res=64)),                      # IQUV through the generic K2
    (-16.0, 32, 0.03, dict(pol=5, freq_res=64, nbit=-32, tscr=4)),
    (16.0, 16, 0.012, dict(pol=5, dm=1.0, coherent=1, freq=316.0)),  # ... through K3 / K4
    (16.0, 32, 0.03, dict(freq_res=64, levels=(-2.75, -0.5, 0.25, 4.5))),
])
def test_extensions_match_the_oracle_on_the_emulator(emu_lib, bw, nchan, secs, kw):
    pu.run_streaming_case(emu_lib, bw, nchan, secs, **kw)


def test_stokes_relations_hold_between_the_two_four_product_modes(emu_lib):
    raw = synth.make_vdif(0.03, bw_mhz=16.0, nchan=32)
    out = {}
    for pol in (4, 5):
        with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.03, pol=pol, nbit=-32, interval=0.0, freq_res=64), emu_lib) as c:
            out[pol] = sigproc.read_fil(c.channelise_bytes(raw)).data.astype(np.float64)
    pp, qq, re, im = (out[4][:, i] for i in range(4))
    i_, q_, u_, v_ = (out[5][:, i] for i in range(4))
    np.testing.assert_allclose(i_, pp + qq, rtol=1e-6)
    np.testing.assert_allclose(v_, pp - qq, rtol=0, atol=1e-6 * np.abs(pp + qq).max())
    np.testing.assert_allclose(q_, 2 * re, rtol=0, atol=1e-6 * np.abs(pp + qq).max())
    np.testing.assert_allclose(u_, 2 * im, rtol=0, atol=1e-6 * np.abs(pp + qq).max())
    assert np.all(i_ * i_ + 1e-3 >= q_ * q_ + u_ * u_ + v_ * v_ - 1e-6 * i_ * i_)     # |polarised| <= I


def test_unpack_tap_all_byte_values_on_the_emulator(emu_lib):
    """decoder 0 (the generic K1's decode) over all 256 byte values, 2-bit and 1-bit, default and custom tables"""
    payload = (np.arange(16000, dtype=np.uint32) * 37 % 256).astype(np.uint8)
    payload[:256] = np.arange(256, dtype=np.uint8)
    for bits, levels in ((2, None), (2, (-2.75, -0.5, 0.25, 4.5)), (1, None)):
        raw = vdif.frame_payload(payload, bw_mhz=32.0, bits=bits)
        nsamp = payload.size * (4 // bits)
        kw = {} if levels is None else {"levels": levels}
        with ch.Channeliser(ch.new_config(emu_lib, bw_mhz=32.0, nchan=64, input_bits=bits, **kw), emu_lib) as c:
            volt = np.zeros((2, nsamp), np.float32)
            c.unpack_device(raw.ctypes.data, 2, 8032, 32, 0, nsamp, 0, volt.ctypes.data, volt.nbytes)
            with pytest.raises(ch.InputError):
                c.unpack_device(raw.ctypes.data, 2, 8032, 32, 0, nsamp + 8, 0, volt.ctypes.data, volt.nbytes)
        lv32 = np.array([-3.3359, -1.0, 1.0, 3.3359], np.float32) if levels is None else levels
        want = (o.unpack_2bit(payload, lv32) if bits == 2 else o.unpack_1bit(payload)).astype(np.float32)
        assert np.array_equal(volt, want), (bits, levels)


def test_push_driven_scan_with_small_blocks_and_an_early_interval(emu_lib, tmp_path):
    """frbch_run_scan without reader / writer threads (flag 1<<22): one 32-MB push is many small batches; a
    -c interval that completes in the first of them must not overflow the scan's row buffer"""
    d = str(tmp_path)
    raws, vd = {}, {}
    for i in (1, 2):
        raws[i] = synth.make_vdif(0.3, bw_mhz=16.0, nchan=32, if_index=i)
        vd[i] = os.path.join(d, f"x_ef_no0001_IF{i}.vdif")
        raws[i].tofile(vd[i])
    chans, parts = [], []
    for i in (2, 1):
        plan = multi_if.plan_ifs(2, 1340.0, 16.0)[i - 1]
        bw = 16.0 if plan.sideband == "u" else -16.0
        chans.append(ch.Channeliser(ch.new_config(emu_lib, bw_mhz=bw, freq_mhz=plan.freq_mhz, nchan=32, freq_res=64, total_s=0.3,
                                                  rescale_constant=1, rescale_interval_s=0.002, max_blocks_per_launch=8,
                                                  flags=1 << 22), emu_lib))
        cfg = o.Config(bw_mhz=bw, freq_mhz=plan.freq_mhz, nchan=32, freq_res=64, total_s=0.3, rescale_interval_s=0.002,
                       source="unknown")
        parts.append(sigproc.read_fil(o.channelise(raws[i], cfg)))
    out = os.path.join(d, "IFall.fil")
    try:
        multi_if.run_scan(chans, [vd[2], vd[1]], out)
    finally:
        for c in chans:
            c.close()
    got = sigproc.read_fil(out)
    want = np.concatenate([p.data for p in parts], axis=2)
    assert got.data.shape == want.shape
    assert np.abs(got.data.astype(int) - want.astype(int)).max() <= 1
    assert np.count_nonzero(got.data != want) <= 1e-4 * want.size


def test_splice_streams_row_blocks_and_matches_numpy_concat(tmp_path):
    """host splice (base2fil.sh:422): block-by-block, all bit widths, unequal lengths, bytes and paths"""
    rng = np.random.default_rng(3)
    for nbits, dt in ((8, np.uint8), (16, "<u2"), (32, "<f4"), (2, None)):
        fils, datas = [], []
        for i, (nch, nt) in enumerate(((16, 37), (8, 41), (16, 40))):
            if nbits == 2:
                codes = rng.integers(0, 4, size=(nt, 2, nch), dtype=np.uint8)
                body = o.pack_codes(codes, 2)
                data = codes
            else:
                data = (rng.integers(0, 200, size=(nt, 2, nch)).astype(dt))
                body = data.tobytes()
            head = o.sigproc_header(telescope="effelsberg", source="x", ra="0:0:0", dec="0:0:0", rawdatafile="x", tstart_mjd=59000.0,
                                    tsamp_s=1e-4, nbits=nbits if nbits != 32 else -32, fch1=1500.0 - 16 * i, foff=-1.0, nchans=nch, nifs=2)
            fils.append(head + body)
            datas.append(data)
        want = np.concatenate([d[:37] for d in datas], axis=2)
        got = sigproc.read_fil(multi_if.splice(fils, block_rows=5))
        assert got.header["nchans"] == 40 and got.header["fch1"] == 1500.0 and got.data.shape == want.shape
        assert np.array_equal(got.data, want)
        paths = []
        for i, blob in enumerate(fils):
            paths.append(str(tmp_path / f"{nbits}_{i}.fil"))
            open(paths[-1], "wb").write(blob)
        out = multi_if.splice(paths, str(tmp_path / f"{nbits}_all.fil"), block_rows=7)
        assert open(out, "rb").read() == multi_if.splice(fils)


def test_bench_launcher_starts_ranks_without_touching_the_gpu():
    """`bench.py --gpus 2` with no WORLD_SIZE: the parent only launches two fresh rank processes (here, without a GPU, every
    rank exits 2 with the loud no-fallback message) and fails the run; it never imports torch or initialises HIP itself"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--no-host", "--seconds", "0.2"], env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert r.returncode == 0 and line["n_gpus"] == 2 and len(line["config"]["per_rank_seconds"]) == 2
    else:
        assert r.returncode != 0 and "rank exit codes [2, 2]" in r.stderr


def test_post_command_line(emu_lib, tmp_path, monkeypatch):
    from frb_baseband_amd import post
    from tests.test_post import HDR, P0, DM0, pulse_train_rows
    monkeypatch.setattr(post._lib, "load", lambda path=None: emu_lib)
    x = pulse_train_rows(6000, HDR)
    fil = str(tmp_path / "a.fil")
    head = o.sigproc_header(telescope="effelsberg", source="J0000+00", ra="0:0:0", dec="0:0:0", rawdatafile="x", tstart_mjd=HDR["tstart"],
                            tsamp_s=HDR["tsamp"], nbits=8, fch1=HDR["fch1"], foff=HDR["foff"], nchans=64, nifs=1)
    open(fil, "wb").write(head + x.tobytes())
    par = tmp_path / "a.par"
    par.write_text("PSRJ J0000+00\nP0 %.6f\nDM %.1f\n" % (P0, DM0))
    assert post.main(["fold", fil, str(par), "--nbin", "64", "-L", "0.2"]) == 0 and os.path.exists(fil + ".ar")
    assert post.main(["prepdata", fil, "--dm", "10", "--dm2", "11", "--clip", "0"]) == 0
    assert os.path.exists(str(tmp_path / "a_DM10.00.dat")) and os.path.exists(str(tmp_path / "a_DM11.00.inf"))


def test_whole_file_path_with_a_mapped_output_equals_the_stream_path(emu_lib, tmp_path, monkeypatch):
    """outputs of 8 MB and more are preallocated and written through a shared mapping by one writer per slot: same
    bytes as push / pull, file length exact, also when rows are emitted interval by interval"""
    raw = synth.make_vdif(0.56, bw_mhz=16.0, nchan=32)
    vd = str(tmp_path / "big.vdif")
    raw.tofile(vd)
    for const, interval in ((1, 10.0), (0, 0.05)):
        cfg = pu.lib_cfg(emu_lib, 16.0, 32, 0.56, freq_res=64, const=const, interval=interval)
        with ch.Channeliser(cfg, emu_lib) as c:
            want = c.channelise_bytes(raw)
        assert len(want) > (8 << 20)
        out = str(tmp_path / f"big_{const}.fil")
        with ch.Channeliser(cfg, emu_lib) as c:
            c.run_file(vd, out)
            assert c.get_info().diag & 1, "the output did not go through the shared mapping (ADVICE r2: O_WRONLY descriptor)"
            c.reset()
            c.run_file(vd, out)                   # again into the existing file (O_TRUNC, then preallocated anew)
            assert c.get_info().diag & 1
        assert open(out, "rb").read() == want
        # no mapping (refused here by the emulator build's test hook): the slot writers take turns in file order
        monkeypatch.setenv("FRBCH_TEST_NO_MMAP", "1")
        with ch.Channeliser(cfg, emu_lib) as c:
            c.run_file(vd, out)
            assert not (c.get_info().diag & 1)
        monkeypatch.delenv("FRBCH_TEST_NO_MMAP")
        assert open(out, "rb").read() == want


def test_bench_maps_profiler_kernel_names_to_timing_slots():
    import sys
    """bench.py matches the rocprofv3 kernel names of its live PMC passes with the engine's timing-slot names
    (frbch_get_timing): every trailing boolean template flag (staging, coherent, statistics) is dropped.  A kernel that
    gains a flag must not silently turn `roofline.traffic` into null."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        import bench
    finally:
        sys.path.remove(root)
    cases = {
        "void fast::frbch_k1_wave<3, 8, 1, true, false>(KParams)": "frbch_k1_wave<3,8,1>",
        "void fast::frbch_k1_wave<4, 8, 2, false, true>(KParams)": "frbch_k1_wave<4,8,2>",
        "void fast::frbch_k2_wave<3, 4, 2, 2, false>(KParams)": "frbch_k2_wave<3,4,2,2>",
        "void fast::frbch_k2_wave<5, 8, 2, 4, true>(KParams)": "frbch_k2_wave<5,8,2,4>",
        "void fast::frbch_k0_stage<4, true>(KParams)": "frbch_k0_stage<4>",
        "void fast::frbch_k2_fast<5, 512>(KParams)": "frbch_k2_fast<5,512>",
        "frbch_quantise": "frbch_quantise",
        "void at::native::vectorized_elementwise_kernel<4>(int)": None,
    }
    for name, want in cases.items():
        assert bench.short_kernel_name(name) == want, name
    # the slot names of the engine for the headline workload, as bench.py looks them up
    assert bench.short_kernel_name("void fast::frbch_k1_wave<3, 8, 1, true, false>(KParams)") in bench.VALU_PER_WAVE_BLOCK


def test_bench_roofline_follows_survey_8d():
    """bench.py: `roofline.frac` is the whole path priced with the SURVEY 8(d) budget (18.5 B per sample for four 8-bit products),
    the dominant kernel is the one with the largest summed launch time WITHOUT exclusions, priced with its share of that budget
    (K1 0.502 + 8, K2 8 + output codes, digitiser / statistics / K0 nothing); the engine's own byte model (float rows of a
    buffered interval included) stays apart"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        import bench
    finally:
        sys.path.remove(root)
    spec = dict(nif=8, bw=32.0, nchan=1024, pol=5, tscrunch=1, seconds=10.0, coherent=False)
    assert abs(bench.budget_bytes_per_sample(spec) - 18.502) < 1e-9
    assert abs(bench.budget_bytes_per_sample(dict(spec, pol=2)) - 17.002) < 1e-9
    assert abs(bench.budget_bytes_per_sample(dict(spec, pol=2, coherent=True)) - 33.002) < 1e-9
    share = bench.budget_share_bytes_per_sample
    assert share("frbch_k1_wave<3,8,1>", spec) == 8.502 and share("frbch_k2_priv<5>", spec) == 10.0
    assert share("frbch_k2_priv<5,stats>", spec) == 10.0            # (a second pass over the spill is priced like the first: no extra credit)
    assert share("frbch_quantise_fast<8>", spec) == 0.0 and share("frbch_k0_stage", spec) == 0.0 and share("frbch_stats", spec) == 0.0
    # the shares of the kernels of one pass add up to the budget
    assert abs(share("frbch_k1_wave<3,8,1>", spec) + share("frbch_k2_wave<3,4,4,2>", spec) - bench.budget_bytes_per_sample(spec)) < 1e-9
    coh = dict(spec, pol=2, coherent=True)
    assert abs(sum(share(k, coh) for k in ("frbch_k1_wave<4,8,2>", "frbch_k2c_fast", "frbch_k3_wave<4>", "frbch_k4_fast")) -
               bench.budget_bytes_per_sample(coh)) < 1e-9
    rec = lambda ms, gb: {"launches": 8, "total_ms": ms, "algorithmic_bytes": gb * 1e9}
    samples = 5.1e9
    timing = {"frbch_k1_wave<3,8,1>": rec(18.2, 43.4), "frbch_k2_wave<3,4,4,2>": rec(15.0, 81.6), "frbch_quantise_fast<8>": rec(18.6, 51.0),
              "frbch_k0_stage": rec(0.9, 5.1)}
    name, r, ach, model = bench.roofline_of(timing, spec, samples)
    assert name == "frbch_quantise_fast<8>" and ach == 0.0 and abs(model - 51.0 / 18.6e-3) < 1e-6      # no exclusions: the digitiser, worth nothing
    del timing["frbch_quantise_fast<8>"]
    name, r, ach, model = bench.roofline_of(timing, spec, samples)
    assert name == "frbch_k1_wave<3,8,1>" and abs(ach - 8.502 * samples / 18.2e-3 / 1e9) < 1e-6
    assert bench.kernels_overlap(timing, 1, 0.030) and not bench.kernels_overlap(timing, 1, 0.0385)
    total, missing = bench.step_traffic({"frbch_k1_wave<3,8,1>": 5.4e9, "frbch_k0_stage": 0.6e9}, timing, 1)
    assert abs(total - 8 * 6.0e9) < 1 and missing == ["frbch_k2_wave<3,4,4,2>"]
