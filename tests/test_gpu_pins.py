"""GPU parity pins that do not need DSPSR ("Next round" item 1):

 * A9/A10 in isolation: the HIP path's own float power and offset/scale pushed through the oracle's rescale +
   digitiser must reproduce the HIP codes with ZERO differing samples (integer work bit-exact);
 * A4 in isolation: the unpack tap decodes all 256 byte values exactly, through every decoder the kernels use;
 * the float stages against the fp64 oracle with the measured error distribution printed (ULP percentiles) and a
   bound of at most twice the measured maximum;
 * BASELINE configs 3 and 4 as stated: 8 IFs x -d4 -t1 -b8 and 2 IFs x 4096 ch -t8 through frbch_run_scan, every
   IF's columns against the oracle.
"""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import multi_if, sigproc, synth
from oracle import frb_oracle as o
from tests import parity_util as pu
from tests.hipmem import DeviceBuffer

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------------------------
# A9 + A10: integer work bit-exact on identical floats
# ------------------------------------------------------------------------------------------------------------------
def _codes_from_hip_floats(power_rows, off, sc, nbit, usb):
    """oracle rescale + digitiser applied to the HIP path's own floats.
    power_rows: float32 [t][nif][chan] in OUTPUT channel order; off/sc: [nif][chan] in INPUT channel order."""
    if usb:
        off, sc = off[:, ::-1], sc[:, ::-1]
    x = o.rescale_apply(power_rows.transpose(1, 2, 0), off, sc)           # [nif][chan][t], fp32 arithmetic
    return o.digitise_values(np.ascontiguousarray(x.transpose(2, 0, 1)), nbit)


def _unpack_codes(buf, nbit, shape):
    flat = sigproc.unpack_samples(buf.tobytes(), 32 if nbit == -32 else nbit)
    return flat.reshape(shape)


@pytest.mark.parametrize("bw,nchan,secs,pol,nbit,tscr,flags", [
    (32.0, 1024, 0.27, 2, 8, 1, 0),            # config 2 kernels: wave K1 + wave K2, statistics fused into K2
    (32.0, 1024, 0.27, 2, 8, 1, 1 << 20),      # ... separate statistics pass
    (-32.0, 1024, 0.27, 4, 8, 1, 0),           # config 3 kernels: four products, MSTAT K2
    (32.0, 1024, 0.27, 4, 16, 2, 0),
    (32.0, 1024, 0.27, 2, 2, 4, 0),
    (32.0, 1024, 0.27, 5, 8, 1, 0),            # IQUV
    (-16.0, 128, 0.05, 2, 16, 1, 0),           # config 1 shape
    (16.0, 128, 0.05, 4, 2, 8, 0),
    (64.0, 4096, 1.1, 2, 8, 8, 0),             # config 4 kernels (M = 32 barrier kernels), -t 8
    (32.0, 1024, 0.27, 2, 8, 1, 3),            # generic kernels
    (32.0, 1024, 0.2, 2, 8, 16, 0),            # frbch_quantise_fast: fewer rows (384) than row phases: only its checked tail runs
    (32.0, 1024, 0.6, 5, 8, 1, 0),             # ... 9 blocks: an even number of pipelined trips + left-over steps
    (32.0, 1024, 0.27, 5, 8, 1, 1 << 28),      # two-pass rescale: both passes of frbch_k2_priv compute the same floats
    (-32.0, 1024, 0.27, 2, 8, 2, 1 << 28),
    (-32.0, 1024, 0.27, 4, 8, 1, 1 << 27),     # the buffered form where two-pass is automatic: float rows by frbch_k2_wave, lean digitiser
    (32.0, 1024, 0.27, 5, 16, 1, 1 << 27),     # ... generic digitiser (16 bit)
])
def test_rescale_and_digitiser_are_bit_exact_on_the_hip_floats(hip_lib, bw, nchan, secs, pol, nbit, tscr, flags):
    raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan)
    d_raw = DeviceBuffer.from_numpy(raw)
    nfr = raw.size // 8032
    cfg = pu.lib_cfg(hip_lib, bw, nchan, secs, pol=pol, nbit=nbit, tscr=tscr, flags=flags)
    with ch.Channeliser(cfg, hip_lib) as c:
        info = c.info
        nblocks = (nfr * 8000) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        ncol = info.nif * nchan
        pw = DeviceBuffer(rows * ncol * 4)
        c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
        power = pw.to_numpy(np.float32).reshape(rows, info.nif, nchan)
        out = DeviceBuffer(rows * info.row_bytes)
        # buffered path: first interval measured, then digitised by frbch_quantise
        r1 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
        r2 = c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
        assert r1 + r2 == rows
        buffered = _unpack_codes(out.to_numpy(np.uint8), nbit, (rows, info.nif, nchan))
        off, sc = c.get_rescale()
        # fused path: K2 digitises in-kernel with the frozen pair
        c.reset()
        c.set_rescale(off, sc)
        r3 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
        assert r3 == rows
        fused = _unpack_codes(out.to_numpy(np.uint8), nbit, (rows, info.nif, nchan))
    want = _codes_from_hip_floats(power, off, sc, nbit, usb=bw > 0)
    assert np.count_nonzero(buffered != want) == 0, "buffered path: codes differ from the oracle digitiser on identical floats"
    assert np.count_nonzero(fused != want) == 0, "fused path: codes differ from the oracle digitiser on identical floats"
    # and the measured offset / scale follow from the same floats (fp64 moments, any summation order)
    p64 = power.astype(np.float64)
    mean = p64.mean(axis=0)
    var = (p64 * p64).mean(axis=0) - mean * mean
    off_out, sc_out = (off[:, ::-1], sc[:, ::-1]) if bw > 0 else (off, sc)
    assert np.abs(off_out + mean).max() <= 4e-7 * np.abs(mean).max()
    np.testing.assert_allclose(sc_out, 1.0 / np.sqrt(var), rtol=2e-6)


@pytest.mark.parametrize("bw,pol,gain", [(32.0, 5, 40.0), (-32.0, 2, 0.02), (32.0, 4, 7.0)])
def test_digitiser_clips_at_both_ends_exactly(hip_lib, bw, pol, gain):
    """the 8-bit digitiser of the fast kernels (one add under round-toward-minus-infinity + v_floor + the saturating
    v_cvt_pk_u8_f32, tools/micro/cvt_probe) with a scale that drives most values into the clips (x 40: Q / U / V far below 0 and
    far above 255) or squeezes them onto a few codes (x 0.02): zero differing codes against the oracle's digitiser on identical floats"""
    nchan, secs = 1024, 0.27
    raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan)
    d_raw = DeviceBuffer.from_numpy(raw)
    nfr = raw.size // 8032
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, secs, pol=pol, nbit=8), hip_lib) as c:
        info = c.info
        nblocks = (nfr * 8000) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        pw = DeviceBuffer(rows * info.nif * nchan * 4)
        c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
        power = pw.to_numpy(np.float32).reshape(rows, info.nif, nchan)
        out = DeviceBuffer(rows * info.row_bytes)
        r1 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
        r1 += c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
        off, sc = c.get_rescale()
        sc = (sc * np.float32(gain)).astype(np.float32)
        c.reset()
        c.set_rescale(off, sc)
        assert c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes) == rows
        got = _unpack_codes(out.to_numpy(np.uint8), 8, (rows, info.nif, nchan))
    want = _codes_from_hip_floats(power, off, sc, 8, usb=bw > 0)
    assert np.count_nonzero(got != want) == 0
    if gain > 1:
        assert (got == 0).mean() > 0.02 and (got == 255).mean() > 0.02          # both clips are exercised


# ------------------------------------------------------------------------------------------------------------------
# A4: every byte value, every decoder
# ------------------------------------------------------------------------------------------------------------------
def _all_bytes_frames(bits):
    """two 8032-byte frames whose payload walks through all 256 byte values (several times, crossing the frame edge)"""
    payload = (np.arange(16000, dtype=np.uint32) * 37 % 256).astype(np.uint8)
    payload[:256] = np.arange(256, dtype=np.uint8)
    from frb_baseband_amd import vdif
    return vdif.frame_payload(payload, bw_mhz=32.0, bits=bits), payload


@pytest.mark.parametrize("levels", [None, (-2.75, -0.5, 0.25, 4.5)])
def test_unpack_tap_all_byte_values(hip_lib, levels):
    raw, payload = _all_bytes_frames(2)
    d_raw = DeviceBuffer.from_numpy(raw)
    nsamp = payload.size * 2
    want = o.unpack_2bit(payload, levels).astype(np.float32)                 # [2][nsamp]
    if levels is None:
        want = o.unpack_2bit(payload, np.array([-3.3359, -1.0, 1.0, 3.3359], np.float32)).astype(np.float32)
    with ch.Channeliser(pu.lib_cfg(hip_lib, 32.0, 1024, 1.0, levels=levels), hip_lib) as c:
        v0 = DeviceBuffer(2 * nsamp * 4)
        c.unpack_device(d_raw.ptr.value, 2, 8032, 32, 0, nsamp, 0, v0.ptr.value, v0.nbytes)
        got0 = v0.to_numpy(np.float32).reshape(2, nsamp)
        v1 = DeviceBuffer(4 * nsamp * 4)
        c.unpack_device(d_raw.ptr.value, 2, 8032, 32, 0, nsamp, 1, v1.ptr.value, v1.nbytes)
        got1 = v1.to_numpy(np.float32).reshape(2, 2, nsamp)
        # an odd start inside the payload: decoder 0 from byte 77 on
        c.unpack_device(d_raw.ptr.value, 2, 8032, 32, 77, nsamp - 154, 0, v0.ptr.value, v0.nbytes)
        got_off = v0.to_numpy(np.float32, count=2 * (nsamp - 154)).reshape(2, nsamp - 154)
    assert set(np.unique(payload)) == set(range(256))
    assert np.array_equal(got0, want)            # generic K1 decode
    assert np.array_equal(got1[0], want)         # frbch_k1_wave: nibble table in the LDS
    assert np.array_equal(got1[1], want)         # frbch_k1_fast: select chain
    assert np.array_equal(got_off, want[:, 154:])


def test_unpack_tap_one_bit(hip_lib):
    raw, payload = _all_bytes_frames(1)
    d_raw = DeviceBuffer.from_numpy(raw)
    nsamp = payload.size * 4
    with ch.Channeliser(ch.new_config(hip_lib, bw_mhz=32.0, nchan=1024, input_bits=1), hip_lib) as c:
        v0 = DeviceBuffer(2 * nsamp * 4)
        c.unpack_device(d_raw.ptr.value, 2, 8032, 32, 0, nsamp, 0, v0.ptr.value, v0.nbytes)
        got = v0.to_numpy(np.float32).reshape(2, nsamp)
    assert np.array_equal(got, o.unpack_1bit(payload).astype(np.float32))


def test_custom_level_table_through_the_whole_path(hip_lib):
    """the level table is data in every kernel family (SURVEY 7 hard part 2): same oracle comparison with another table"""
    lv = (-2.75, -0.5, 0.25, 4.5)
    pu.run_streaming_case(hip_lib, 32.0, 1024, 0.14, levels=lv)                      # wave K1 (nibble table)
    pu.run_streaming_case(hip_lib, -32.0, 1024, 0.3, levels=lv, dm=56.7, coherent=1, freq=400.0, tscr=4)   # barrier K1 of the coherent path (select chain)
    pu.run_streaming_case(hip_lib, -16.0, 128, 0.05, levels=lv, flags=1, pol=4)      # generic K1


# ------------------------------------------------------------------------------------------------------------------
# float stages: measured error distribution against the fp64 oracle
# ------------------------------------------------------------------------------------------------------------------
def _ulp32(x):
    return np.spacing(np.abs(x).astype(np.float32)).astype(np.float64)


POWER_CASES = [
    # bw, nchan, secs, pol, tscr, kwargs (N = 2*nchan*freq_res samples per block)
    ("cfg1 128ch N=2^17", 16.0, 128, 0.05, 2, 1, {}),
    ("cfg2 1024ch N=2^22", 32.0, 1024, 0.14, 2, 1, {}),
    ("cfg3 1024ch -d4", -32.0, 1024, 0.14, 4, 1, {}),
    ("iquv 1024ch", 32.0, 1024, 0.14, 5, 1, {}),
    ("2048ch N=2^24", 64.0, 2048, 0.3, 2, 1, {}),
    ("cfg4 4096ch N=2^26 -t8", 64.0, 4096, 1.1, 2, 8, {}),
    ("cfg5 2048ch -D56.7 coherent", 32.0, 2048, 0.6, 2, 1, dict(dm=56.7, coherent=1, freq=1400.0, freq_res=4096)),
]


@pytest.mark.parametrize("name,bw,nchan,secs,pol,tscr,kw", POWER_CASES, ids=[c[0] for c in POWER_CASES])
def test_power_error_distribution(hip_lib, name, bw, nchan, secs, pol, tscr, kw):
    """|P - P_oracle| per sample in fp32 ULPs of the oracle value (median / 99.9 % / max) and relative to the channel's
    mean power; the asserted bound is parity_util.POWER_RTOL (= at most 2x the largest maximum measured over these
    configurations).  1 ULP against an fp64 oracle is not attainable for fp32 FFTs of 2^17..2^26
    points: the table printed here is what IS attained."""
    raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan)
    ocfg = pu.oracle_cfg(bw, nchan, secs, pol=pol, tscr=tscr, **kw)
    o.channelise(raw, ocfg)
    want = ocfg.result["power"]                                            # [nif][C][nt]
    if bw > 0:
        want = want[:, ::-1, :]
    want = want.transpose(2, 0, 1)
    d_raw = DeviceBuffer.from_numpy(raw)
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, secs, pol=pol, tscr=tscr, **kw), hip_lib) as c:
        info = c.info
        nfr = raw.size // 8032
        nblocks = (nfr * 8000 - info.block_payload_bytes) // info.block_stride_bytes + 1
        pw = DeviceBuffer(nblocks * info.rows_per_block * info.nif * nchan * 4)
        c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
        got = pw.to_numpy(np.float32).reshape(want.shape).astype(np.float64)
    err = np.abs(got - want)
    # reference magnitude of a (product, channel): mean |value| of products that change sign (Q, U, V, Re/Im PQ*) is the
    # mean total power of the channel
    tot = want[:, :1, :] if pol != 4 else want[:, 0:1, :] + want[:, 1:2, :]
    chan_scale = np.abs(tot).mean(axis=0, keepdims=True)
    rel = (err / chan_scale).max()
    ulps = err / _ulp32(want)
    stats = {"case": name, "samples": int(err.size), "rel_to_channel_mean_max": float(rel),
             "ulp_median": float(np.median(ulps)), "ulp_p999": float(np.quantile(ulps, 0.999)), "ulp_max": float(ulps.max())}
    # The same chain in fp32 on the CPU (oracle/frb_oracle.c: plain-C Stockham FFTs, the closest thing here to what the reference
    # really launches -- digifil is an fp32 CPU program, process_vdif.py:157-161) against the same fp64 oracle.  north_star's
    # "within 1 ULP of the digifil path" cannot be tested without DSPSR; what CAN be stated is how the HIP chain's error relates to
    # that of an fp32 CPU implementation of the same transform.  Measured on MI355X (round 4, profiles/r04_power_error_vs_cport.jsonl):
    # the HIP chain's 99.9 % point is 1.5 - 2.1 x and its maximum 1.5 - 2.0 x the C port's -- same order, not equal: the C port takes
    # every twiddle from an fp64-evaluated table, the register kernels compose theirs from a few base factors per lane (one or two
    # more roundings per twiddle) and carry the fractional-delay products besides.  Asserted: within 2.5 x, both.
    if not kw.get("coherent"):
        from oracle import c_oracle
        r_ = ocfg.result["geometry"][0]
        payload = raw[: raw.size // 8032 * 8032].reshape(-1, 8032)[:, 32:].reshape(-1)
        cp = c_oracle.block_power(payload, nchan, r_, nblocks, pol, tscr).astype(np.float64)      # [nif][C][nt], ascending channels
        if bw > 0:
            cp = cp[:, ::-1, :]
        cerr = np.abs(cp.transpose(2, 0, 1) - want)
        c_rel = cerr / chan_scale
        h_rel = err / chan_scale
        stats.update({"cport_rel_max": float(c_rel.max()), "cport_rel_p999": float(np.quantile(c_rel, 0.999)),
                      "hip_rel_p999": float(np.quantile(h_rel, 0.999)),
                      "cport_ulp_median": float(np.median(cerr / _ulp32(want))), "cport_ulp_p999": float(np.quantile(cerr / _ulp32(want), 0.999)),
                      "cport_ulp_max": float((cerr / _ulp32(want)).max())})
    print("POWER-ERR " + json.dumps(stats))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/power_error_distribution.jsonl", "a") as f:
        f.write(json.dumps(stats) + "\n")
    bound = pu.power_rtol(nchan, ocfg.result["geometry"][0], tscr)       # regression guard: 2 x the fp32-chain error model
    stats["bound"] = bound
    assert rel <= bound, stats
    if "cport_rel_max" in stats:
        assert stats["rel_to_channel_mean_max"] <= 2.5 * stats["cport_rel_max"], stats
        assert stats["hip_rel_p999"] <= 2.5 * stats["cport_rel_p999"], stats


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs 3 and 4 as stated, through frbch_run_scan
# ------------------------------------------------------------------------------------------------------------------
def _scan_vs_oracle(tmp_path, nif, bw, nchan, secs, pol, tscr):
    d = str(tmp_path)
    raws, vd = {}, {}
    for i in range(1, nif + 1):
        raws[i] = synth.make_vdif(secs, bw_mhz=bw, nchan=nchan, if_index=i)
        vd[i] = os.path.join(d, f"x_ef_no0001_IF{i}.vdif")
        raws[i].tofile(vd[i])
    kw = dict(freq_lsb_0=1340.0, bw=bw, nchan=nchan, nsec=secs, pol=pol, tscrunch=tscr, source="R3", ra="01:58:00.75",
              dec="65:43:00.3")
    with contextlib.redirect_stdout(io.StringIO()):
        out = multi_if.process_scan(vd, out_dir=d, direct=True, **kw)
    got = sigproc.read_fil(out)
    nprod = 4 if pol >= 4 else 1
    assert got.header["nchans"] == nif * nchan and got.data.shape[1:] == (nprod, nif * nchan)
    plans = {p.index: p for p in multi_if.plan_ifs(nif, 1340.0, bw)}
    total_bad = 0
    for col, i in enumerate(multi_if.splice_order(nif)):               # highest IF first (base2fil.sh:350,367)
        plan = plans[i]
        cfg = o.Config(bw_mhz=bw if plan.sideband == "u" else -bw, freq_mhz=plan.freq_mhz, nchan=nchan, total_s=secs,
                       pol_mode=pol, tscrunch=tscr, source="R3", ra="01:58:00.75", dec="65:43:00.3")
        ref = o.channelise(raws[i], cfg)
        want = sigproc.read_fil(ref).data
        mine = got.data[:, :, col * nchan:(col + 1) * nchan]
        assert mine.shape == want.shape, (mine.shape, want.shape)
        total_bad += pu.check_code_arrays(want, mine, cfg)              # identical except at rounding ties of the oracle value
        if col == 0:
            assert got.header["fch1"] == pytest.approx(sigproc.read_fil(ref).header["fch1"])
    return total_bad


def test_config3_as_stated_8_ifs_d4_t1_8bit(tmp_path):
    """BASELINE.json configs[2]: 8 IFs x 32 MHz -> 1024 ch, -d4, tscrunch 1, 8 bit, ONE GPU, one IFall file; every IF's
    columns (all four products) against the oracle.  2 blocks per IF."""
    _scan_vs_oracle(tmp_path, 8, 32.0, 1024, 0.14, 4, 1)


def test_config3_iquv_8_ifs(tmp_path):
    """the IQUV spelling of the same configuration (north_star "full-Stokes IQUV"): 8 IFs, as stated"""
    _scan_vs_oracle(tmp_path, 8, 32.0, 1024, 0.14, 5, 1)


def test_config4_share_two_4096ch_ifs_one_gpu(tmp_path):
    """BASELINE.json configs[3], one GPU's share: 2 IFs x 64 MHz -> 4096 ch Stokes I, -t 8 (the golden command line
    `-t 8 ... -F4096:8192`), through frbch_run_scan.  2 blocks of 2^26 samples per IF."""
    _scan_vs_oracle(tmp_path, 2, 64.0, 4096, 1.1, 2, 8)


# ------------------------------------------------------------------------------------------------------------------
# handle / device hygiene
# ------------------------------------------------------------------------------------------------------------------
def test_handle_driven_from_another_thread(hip_lib, tmp_path):
    """every entry point sets the handle's device itself: open here, run_file + get_rescale from a fresh thread"""
    import threading
    raw = synth.make_vdif(0.05, bw_mhz=16.0, nchan=128)
    vd = str(tmp_path / "a.vdif")
    raw.tofile(vd)
    res = {}
    with ch.Channeliser(pu.lib_cfg(hip_lib, 16.0, 128, 0.05), hip_lib) as c:
        def work():
            try:
                c.run_file(vd, str(tmp_path / "a.fil"))
                res["resc"] = c.get_rescale()
            except Exception as exc:   # noqa: BLE001
                res["exc"] = exc
        th = threading.Thread(target=work)
        th.start()
        th.join(timeout=120)
    assert "exc" not in res, res.get("exc")
    ocfg = pu.oracle_cfg(16.0, 128, 0.05)
    pu.check_codes(o.channelise(raw, ocfg), open(str(tmp_path / "a.fil"), "rb").read(), ocfg)


def test_reset_waits_for_the_callers_stream(hip_lib):
    """frbch_reset / get_rescale after asynchronous work on a caller stream (bench.py's step loop): results equal the
    synchronous sequence, run after run"""
    torch = pytest.importorskip("torch")
    raw = synth.make_vdif(1.0, bw_mhz=32.0, nchan=1024)
    frames = torch.from_numpy(raw).cuda()
    nfr = raw.size // 8032
    side = torch.cuda.Stream()
    with ch.Channeliser(pu.lib_cfg(hip_lib, 32.0, 1024, 1.0), hip_lib) as c:
        info = c.info
        nblocks = (nfr * 8000) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        out = torch.empty(rows * info.row_bytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        results = []
        for _ in range(4):
            c.reset()                                                   # must wait for the previous trip's stream
            r1 = c.process_device(frames.data_ptr(), nfr, 8032, 32, 0, nblocks, out.data_ptr(), out.numel(), side.cuda_stream)
            r2 = c.flush_device(out.data_ptr() + r1 * info.row_bytes, out.numel() - r1 * info.row_bytes, side.cuda_stream)
            assert r1 + r2 == rows
            off, sc = c.get_rescale()                                   # no explicit synchronisation by the caller
            side.synchronize()
            results.append((off.copy(), sc.copy(), out.cpu().numpy().copy()))
    for off, sc, codes in results[1:]:
        assert np.array_equal(off, results[0][0]) and np.array_equal(sc, results[0][1])
        assert np.array_equal(codes, results[0][2])


# ------------------------------------------------------------------------------------------------------------------
# invalid frames and missing frames (extract_baseband_chunk.py:56-69 reads the same header fields)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pol,via_file", [(2, False), (4, True), (2, True)])
def test_dropped_and_invalid_frames(hip_lib, tmp_path, pol, via_file):
    """config-2 shape, 4 blocks: three frames dropped inside block 1 (the stream path fills them with zero frames so
    that every later sample keeps its time), one frame flagged invalid inside block 2 (read as zero voltages); blocks
    0 and 3 are clean, the touched blocks run the SAME wave K1 in its masked form (a flag per n2 row beside the staged
    payload; round 3 sent them to the generic radix-2 K1): no generic kernel appears in the timing report"""
    raw = synth.make_vdif(0.27, bw_mhz=32.0, nchan=1024)
    nfr = raw.size // 8032
    fr = raw.reshape(nfr, 8032).copy()
    fr[700, 3] |= 0x80                                             # block 2 (frames 524..786): invalid
    keep = np.ones(nfr, bool)
    keep[[300, 301, 302]] = False                                  # block 1 (frames 262..524): three frames missing
    hurt = fr[keep].reshape(-1)
    ocfg = pu.oracle_cfg(32.0, 1024, 0.27, pol=pol)
    ref = o.channelise(hurt, ocfg)
    assert ocfg.result["frame_counters"] == dict(gaps=1, filled=3, invalid=1)
    cfg = pu.lib_cfg(hip_lib, 32.0, 1024, 0.27, pol=pol)
    with ch.Channeliser(cfg, hip_lib) as c:
        c.set_profiling(True)
        if via_file:
            vd = str(tmp_path / "hurt.vdif")
            hurt.tofile(vd)
            c.run_file(vd, str(tmp_path / "hurt.fil"))
            got = open(str(tmp_path / "hurt.fil"), "rb").read()
        else:
            got = c.channelise_bytes(hurt)
        info = c.get_info()
        names = {k for k, v in c.get_timing().items() if v["launches"]}
    # (a launch that falls back to the generic K1 renames the slot "<planned>+frbch_k1_branch")
    assert "frbch_k1_wave<3,8,1>" in names and "frbch_k0_stage" in names and not any("frbch_k1_branch" in n for n in names), names
    assert (info.frames_invalid, info.frame_gaps, info.frames_filled) == (1, 1, 3)
    pu.check_codes(ref, got, ocfg)
    # only invalid flags (no gap): the overlapped whole-file path keeps running and masks the frame
    inv = fr.reshape(-1)
    ocfg2 = pu.oracle_cfg(32.0, 1024, 0.27, pol=pol)
    ref2 = o.channelise(inv, ocfg2)
    vd = str(tmp_path / "inv.vdif")
    inv.tofile(vd)
    with ch.Channeliser(cfg, hip_lib) as c:
        c.run_file(vd, str(tmp_path / "inv.fil"))
        assert c.get_info().frames_invalid == 1 and c.get_info().frames_filled == 0
    pu.check_codes(ref2, open(str(tmp_path / "inv.fil"), "rb").read(), ocfg2)


def test_invalid_frame_at_4096_channels(hip_lib):
    """config-4 kernels, 2 blocks of 2^26 samples: block 0 is clean, block 1 holds an invalid frame -- since round 4 the SAME wave
    K1 runs it in its masked form (R = 8192: one flag BIT per n2 row beside the staged payload; round 3: the generic K1)"""
    raw = synth.make_vdif(1.1, bw_mhz=64.0, nchan=4096)
    nfr = raw.size // 8032
    fr = raw.reshape(nfr, 8032).copy()
    fr[6000, 3] |= 0x80                                            # block 1 (frames 4194..8388)
    inv = fr.reshape(-1)
    ocfg = pu.oracle_cfg(64.0, 4096, 1.1)
    ref = o.channelise(inv, ocfg)
    assert ocfg.result["frame_counters"]["invalid"] == 1
    with ch.Channeliser(pu.lib_cfg(hip_lib, 64.0, 4096, 1.1), hip_lib) as c:
        c.set_profiling(True)
        got = c.channelise_bytes(inv)
        assert c.get_info().frames_invalid == 1
        names = {k for k, v in c.get_timing().items() if v["launches"]}
    assert "frbch_k1_wave<5,8,4>" in names and not any("frbch_k1_branch" in n for n in names), names
    pu.check_codes(ref, got, ocfg)


# ------------------------------------------------------------------------------------------------------------------
# Input variants through the FAST kernels: legacy 16-byte headers and 1000-byte payloads
# (mode VDIF_1000-1024-16-2, spif2file.sh:48-52; header size logic base2fil.sh:137-147)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("payload,legacy", [(1000, 1), (8000, 1), (1000, 0)])
def test_legacy_headers_and_small_payloads_through_the_fast_kernels(hip_lib, payload, legacy):
    bw, nchan, secs = 32.0, 1024, 0.14
    raw = synth.make_vdif(secs, bw_mhz=bw, nchan=nchan, payload_bytes=payload, legacy=legacy)
    ocfg = pu.oracle_cfg(bw, nchan, secs)
    ref = o.channelise(raw, ocfg)
    hb = 16 if legacy else 32
    fb = payload + hb
    # host stream path (frbch_push: geometry from the first frame header) ...
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, secs), hip_lib) as c:
        c.set_profiling(True)
        got = c.channelise_bytes(raw)
        names = {k for k, v in c.get_timing().items() if v["launches"]}
        info = c.get_info()
    pu.check_codes(ref, got, ocfg)
    assert info.frame_bytes == fb and info.header_bytes == hb
    assert any(n.startswith("frbch_k1_wave") for n in names) and "frbch_k0_stage" in names, names   # not the generic K1
    assert any(n.startswith("frbch_k2_wave") or n.startswith("frbch_k2_priv") for n in names), names
    # ... and the device-resident entry point with the frame geometry handed over
    d_raw = DeviceBuffer.from_numpy(raw)
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, secs), hip_lib) as c:
        info = c.info
        nfr = raw.size // fb
        nblocks = (nfr * payload) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        out = DeviceBuffer(rows * info.row_bytes)
        r1 = c.process_device(d_raw.ptr.value, nfr, fb, hb, 0, nblocks, out.ptr.value, out.nbytes)
        r2 = c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
        assert r1 + r2 == rows
        codes = out.to_numpy(np.uint8).reshape(rows, 1, nchan)
    pu.check_code_arrays(sigproc.read_fil(ref).data[:rows], codes, ocfg)


# ------------------------------------------------------------------------------------------------------------------
# frbch_scan_device: the scan with everything in HBM (the bench's step), rows of all IFs in one row buffer
# ------------------------------------------------------------------------------------------------------------------
def _scan_device(hip_lib, nif, bw, nchan, secs, overlap=0, **kw):
    raws = [synth.make_vdif(secs, bw_mhz=bw, nchan=nchan, if_index=i + 1) for i in range(nif)]
    bufs = [DeviceBuffer.from_numpy(r) for r in raws]
    chans, ocfgs = [], []
    for i in range(nif):
        sbw = bw if i % 2 else -bw
        cfg = pu.lib_cfg(hip_lib, sbw, nchan, secs, **kw)
        cfg.overlap = overlap
        chans.append(ch.Channeliser(cfg, hip_lib))
        ocfgs.append(pu.oracle_cfg(sbw, nchan, secs, **{k: v for k, v in kw.items() if k not in ("maxb", "flags")}))
    info = chans[0].info
    nfr = raws[0].size // 8032
    nblocks = (nfr * 8000) // info.block_payload_bytes
    rows = nblocks * info.rows_per_block
    out = DeviceBuffer(rows * nif * info.row_bytes)
    got = multi_if.scan_device(chans, [b.ptr.value for b in bufs], nfr, 8032, 32, 0, nblocks, out.ptr.value, rows)
    assert got == rows
    data = out.to_numpy(np.uint8 if kw.get("nbit", 8) == 8 else np.uint16).reshape(rows, info.nif, nif * nchan)
    for c in chans:
        c.close()
    return raws, ocfgs, data, rows


def test_scan_device_config3_as_stated_against_the_oracle(hip_lib):
    """BASELINE configs[2] as the bench runs it: 8 IFs x 32 MHz -> 1024-ch IQUV, 8 bit, one call, one row buffer; every
    IF's columns (all four products) against the oracle.  2 blocks per IF."""
    raws, ocfgs, data, rows = _scan_device(hip_lib, 8, 32.0, 1024, 0.14, pol=5)
    for i, (raw, ocfg) in enumerate(zip(raws, ocfgs)):
        want = sigproc.read_fil(o.channelise(raw, ocfg)).data[:rows]
        pu.check_code_arrays(want, data[:, :, i * 1024:(i + 1) * 1024], ocfg)


@pytest.mark.parametrize("overlap,kw", [
    (192 | (3 << 24), dict(pol=5, flags=1 << 27)),                             # the digitiser beside the next IF's K1, its CUs held by an LDS reservation (buffered form forced)
    (160 | (3 << 24), dict(pol=2, interval=0.1, const=0, maxb=2)),             # ... an interval per 0.1 s (the power buffer is re-used)
    (0, dict(pol=5, interval=0.1, maxb=2, flags=1 << 27)),                     # the automatic overlap, the interval ends inside the scan
    (0, dict(pol=5)),                                                          # the automatic rescale form (two-pass: no digitiser to overlap)
])
def test_scan_device_lanes_give_the_same_rows(hip_lib, overlap, kw):
    """two kernels sharing the chip changes WHERE and WHEN kernels run, never what they write: rows identical to the run without
    overlap (frbch_config.overlap = 1), bit for bit"""
    _r, _o, plain, rows = _scan_device(hip_lib, 3, 32.0, 1024, 0.27, overlap=1, **kw)
    _r, _o, lanes, rows2 = _scan_device(hip_lib, 3, 32.0, 1024, 0.27, overlap=overlap, **kw)
    assert rows == rows2 and np.array_equal(plain, lanes)


def test_scan_device_coherent_filterbank_rows_equal_the_per_if_rows(hip_lib):
    """the known-pulsar flags (-D <dm> -F C:D, process_vdif.py:177-180) on several IFs of one GPU: K4 (frbch_k4_fast) writes every
    IF's rows into its columns of the scan's row buffer -- bit-identical to the IF channelised alone"""
    kw = dict(dm=26.7, coherent=1, freq=350.0, pol=5, tscr=2)
    bw, nchan, secs, nif = 32.0, 512, 0.15, 3
    raws = [synth.make_vdif(secs, bw_mhz=bw, nchan=nchan, if_index=i + 1) for i in range(nif)]
    bufs = [DeviceBuffer.from_numpy(r) for r in raws]
    chans = [ch.Channeliser(pu.lib_cfg(hip_lib, -bw if i % 2 else bw, nchan, secs, **kw), hip_lib) for i in range(nif)]
    info = chans[0].info
    nfr = raws[0].size // 8032
    nblocks = (nfr * 8000 - info.block_payload_bytes) // info.block_stride_bytes + 1
    rows = nblocks * info.rows_per_block
    out = DeviceBuffer(rows * nif * info.row_bytes)
    assert multi_if.scan_device(chans, [b.ptr.value for b in bufs], nfr, 8032, 32, 0, nblocks, out.ptr.value, rows) == rows
    scan = out.to_numpy(np.uint8).reshape(rows, 4, nif * nchan)
    single = DeviceBuffer(rows * info.row_bytes)
    for i, c in enumerate(chans):
        c.reset()
        r1 = c.process_device(bufs[i].ptr.value, nfr, 8032, 32, 0, nblocks, single.ptr.value, single.nbytes)
        r1 += c.flush_device(single.ptr.value + r1 * info.row_bytes, single.nbytes - r1 * info.row_bytes)
        assert r1 == rows
        assert np.array_equal(single.to_numpy(np.uint8).reshape(rows, 4, nchan), scan[:, :, i * nchan:(i + 1) * nchan])
        c.close()
