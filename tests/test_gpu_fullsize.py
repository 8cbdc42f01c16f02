"""GPU tests at BASELINE.json's full size (configs[1]: 10 s of a 32 MHz IF = 152 blocks of 2^22 samples)
through size-independent properties, plus the end-to-end drop-in flows (shim into a FIFO, multi-IF scan)."""
import contextlib
import io
import os
import threading

import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import multi_if, process_vdif as pv, sigproc, synth
from oracle import frb_oracle as o
from tests import parity_util as pu
from tests.hipmem import DeviceBuffer

pytestmark = pytest.mark.gpu

LEVEL = np.array([-3.3359, -1.0, 1.0, 3.3359])


@pytest.fixture(scope="module")
def ten_seconds():
    return synth.make_vdif(10.0, bw_mhz=32.0, nchan=1024)


def test_parseval_every_block_full_size(hip_lib, ten_seconds):
    """sum_k sum_t (|pol0_k[t]|^2 + |pol1_k[t]|^2) of a block = R * (half-spectrum energy), and the
    half-spectrum energy follows exactly from the time samples:
    sum_{m<N/2} |X[m]|^2 = (N * sum x^2 + X[0]^2 - X[N/2]^2) / 2  with X[0] = sum x, X[N/2] = sum (-1)^n x."""
    raw = ten_seconds
    bw, nchan, r = 32.0, 1024, 2048
    n = 2 * nchan * r
    payload = o.strip_frames(raw, 8032, 32)
    nblocks = payload.size * 2 // n
    assert nblocks == 152
    d_raw = DeviceBuffer.from_numpy(raw)
    nfr = raw.size // 8032
    got = np.empty(nblocks)
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, 10.0), hip_lib) as c:
        info = c.info
        step = 19
        pw = DeviceBuffer(step * info.rows_per_block * nchan * 4)
        for b0 in range(0, nblocks, step):
            nb = min(step, nblocks - b0)
            c.power_device(d_raw.ptr.value, nfr, 8032, 32, b0 * info.block_payload_bytes, nb, pw.ptr.value, pw.nbytes)
            p = pw.to_numpy(np.float32, count=nb * info.rows_per_block * nchan).reshape(nb, -1)
            got[b0:b0 + nb] = p.astype(np.float64).sum(axis=1)
    want = np.empty(nblocks)
    for b in range(nblocks):
        x = o.unpack_2bit(payload[b * n // 2:(b + 1) * n // 2])          # [2][N]
        e = 0.0
        for pol in range(2):
            xs = x[pol]
            alt = xs[0::2].sum() - xs[1::2].sum()
            e += (n * (xs * xs).sum() + xs.sum() ** 2 - alt ** 2) / 2.0
        want[b] = r * e
    np.testing.assert_allclose(got, want, rtol=2e-6)


def test_full_size_is_deterministic_and_rescale_is_reusable(hip_lib, ten_seconds):
    raw = ten_seconds
    cfg = pu.lib_cfg(hip_lib, 32.0, 1024, 10.0)
    d_raw = DeviceBuffer.from_numpy(raw)
    nfr = raw.size // 8032
    outs = []
    with ch.Channeliser(cfg, hip_lib) as c:
        info = c.info
        nblocks = (nfr * 8000) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        out = DeviceBuffer(rows * info.row_bytes)
        for _ in range(2):
            c.reset()
            r1 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
            r2 = c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
            assert r1 + r2 == rows
            outs.append(out.to_numpy(np.uint8))
        off, sc = c.get_rescale()
        c.reset()
        c.set_rescale(off, sc)                     # fused path (K2 digitises) with the measured scale
        r3 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
        assert r3 == rows
        fused = out.to_numpy(np.uint8)
    assert np.array_equal(outs[0], outs[1])        # run-to-run bit identical
    assert np.array_equal(outs[0], fused)          # buffered first-interval path == fused path, same scale
    d = outs[0].astype(np.float64)
    assert abs(d.mean() - 127.5) < 0.6 and abs(d.std() - 127.5 / 6) < 1.5


def test_statistics_summed_by_k2_equal_the_separate_pass(hip_lib, ten_seconds):
    """A9 fused into K2 (fp32 partial sums per workgroup, fp64 across workgroups) against the separate fp64 pass over
    the power buffer (flag bit 20), for an interval that is the whole file and for one that ends inside a block."""
    raw = ten_seconds[: 8032 * 4000 * 2]                      # 2 s
    d_raw = DeviceBuffer.from_numpy(raw)
    nfr = raw.size // 8032
    for interval, pol in ((10.0, 2), (0.7, 2), (10.0, 4), (0.7, 4)):     # pol 4: a K2 thread owns four column groups
        res = []
        for flags in (0, 1 << 20):
            cfg = pu.lib_cfg(hip_lib, 32.0, 1024, 2.0, interval=interval, flags=flags, pol=pol)
            with ch.Channeliser(cfg, hip_lib) as c:
                info = c.info
                nblocks = (nfr * 8000) // info.block_payload_bytes
                out = DeviceBuffer(nblocks * info.rows_per_block * info.row_bytes)
                r1 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
                r2 = c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
                assert r1 + r2 == nblocks * info.rows_per_block
                res.append((c.get_rescale(), out.to_numpy(np.uint8)))
        (off_a, sc_a), codes_a = res[0]
        (off_b, sc_b), codes_b = res[1]
        assert np.abs(off_a - off_b).max() <= 2e-7 * np.abs(off_b).max()
        np.testing.assert_allclose(sc_a, sc_b, rtol=1e-6)
        diff = codes_a.astype(np.int16) - codes_b.astype(np.int16)
        assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) < 1e-4 * diff.size


def test_digifil_shim_into_fifo(tmp_path):
    """the flag-compatible `digifil` executable, driven through the harness mirror, writing into a pre-made FIFO
    (base2fil.sh:348-349), read concurrently like `splice` would."""
    raw = synth.make_vdif(0.14, bw_mhz=32.0, nchan=1024)
    vd = str(tmp_path / "pr001a_ef_no0001_IF1.vdif")
    raw.tofile(vd)
    hdr = pv.make_hdr("R3", 1340.49, vd, pol=2, usb=False, ra="01:58:00.7502", dec="65:43:00.3152", bw=32.0,
                      telescope="effelsberg")
    fifodir = tmp_path / "fifos"
    fifodir.mkdir()
    fifo = str(fifodir / (os.path.basename(vd) + "_pol2.fil"))
    os.mkfifo(fifo)
    got = {}

    def reader():
        with open(fifo, "rb") as f:
            got["data"] = f.read()
    th = threading.Thread(target=reader)
    th.start()
    with contextlib.redirect_stdout(io.StringIO()):
        ret = pv.run_digifil(hdr, str(fifodir), 0, 0.14, 1024, overwrite=True, pol=2, nbit=8, backend="shim")
    th.join(timeout=60)
    assert ret == str(fifodir) + "/" + os.path.basename(fifo) and os.path.exists(fifo)
    ocfg = o.config_from_hdr(hdr, nchan=1024, total_s=0.14)
    pu.check_codes(o.channelise(raw, ocfg), got["data"], ocfg)
    hdr_fields = sigproc.read_fil(got["data"]).header
    assert hdr_fields["telescope_id"] == 8 and hdr_fields["foff"] < 0 and hdr_fields["source_name"] == "R3"
    with pytest.raises(pv.RunError), contextlib.redirect_stdout(io.StringIO()):
        pv.run_digifil(str(tmp_path / "missing.hdr"), str(tmp_path), 0, 1, 1024, overwrite=True, backend="shim")


def test_fifo_sink_takes_the_pinned_ring_by_reference(hip_lib, tmp_path, monkeypatch):
    """base2fil.sh:348-350: the sink of a scan is a named pipe.  Outputs of more than two pipes' worth per slot go out through
    vmsplice of the hipHostMalloc'ed ring (frbch_info.diag bit 1) -- same bytes as a regular file and as with FRBCH_FIFO_COPY=1"""
    raw = synth.make_vdif(1.0, bw_mhz=32.0, nchan=1024)
    vd = str(tmp_path / "a_IF1.vdif")
    raw.tofile(vd)
    kw = dict(pol=5, interval=0.3)                     # 4096-byte rows, 31 k rows: 128 MB
    out = str(tmp_path / "o.fil")
    with ch.Channeliser(pu.lib_cfg(hip_lib, 32.0, 1024, 1.0, **kw), hip_lib) as c:
        c.run_file(vd, out)
    want = open(out, "rb").read()
    assert len(want) > 64 << 20
    for copy in (False, True):
        if copy:
            monkeypatch.setenv("FRBCH_FIFO_COPY", "1")
        fifo = str(tmp_path / f"f{int(copy)}.fil")
        os.mkfifo(fifo)
        got = {}

        def reader():
            with open(fifo, "rb") as f:
                got["data"] = f.read()
        th = threading.Thread(target=reader)
        th.start()
        with ch.Channeliser(pu.lib_cfg(hip_lib, 32.0, 1024, 1.0, **kw), hip_lib) as c:
            c.run_file(vd, fifo)
            diag = c.get_info().diag
        th.join(timeout=120)
        assert got["data"] == want
        assert bool(diag & 2) == (not copy)


def test_known_pulsar_flags_through_the_harness(tmp_path):
    """process_vdif.py:177-180: `-D <dm> -F<C>:D` appended after `-F<C>:<R>` (last wins) -> coherent filterbank,
    in-process through the C-ABI; refdm lands in the SIGPROC header."""
    raw = synth.make_vdif(0.3, bw_mhz=32.0, nchan=1024)
    vd = str(tmp_path / "pr001a_ef_no0002_IF1.vdif")
    raw.tofile(vd)
    hdr = pv.make_hdr("B0329+54", 420.0, vd, pol=2, usb=True, ra="03:32:59.4", dec="54:34:43.3", bw=32.0,
                      telescope="effelsberg")
    with contextlib.redirect_stdout(io.StringIO()) as log:
        fil = pv.run_digifil(hdr, str(tmp_path), 0, 0.3, 1024, overwrite=True, pol=2, nbit=8, tscrunch=4, dm=26.7,
                             coherent=True)
    assert "-D 26.7 -F1024:D" in log.getvalue()
    got = open(fil, "rb").read()
    ocfg = o.config_from_hdr(hdr, nchan=1024, total_s=0.3, tscrunch=4, dm=26.7, coherent=True)
    pu.check_codes(o.channelise(raw, ocfg), got, ocfg)
    r, pos, neg, keep = ocfg.result["geometry"]
    assert r == 2048 and pos > 0 and neg > 0 and keep % 4 == 0
    assert sigproc.read_fil(got).header["refdm"] == 26.7


def test_multi_if_scan_on_device(tmp_path):
    d = str(tmp_path)
    raws, vd = {}, {}
    for i in (1, 2):
        raws[i] = synth.make_vdif(0.14, bw_mhz=32.0, nchan=1024, if_index=i)
        vd[i] = os.path.join(d, f"x_ef_no0001_IF{i}.vdif")
        raws[i].tofile(vd[i])
    with contextlib.redirect_stdout(io.StringIO()):
        out = multi_if.process_scan(vd, freq_lsb_0=1340.0, bw=32.0, nchan=1024, nsec=0.14, out_dir=d, source="R3",
                                    ra="01:58:00.75", dec="65:43:00.3")
    got = sigproc.read_fil(out)
    parts, ocfgs = [], []
    for i in (2, 1):
        plan = multi_if.plan_ifs(2, 1340.0, 32.0)[i - 1]
        cfg = o.Config(bw_mhz=32.0 if plan.sideband == "u" else -32.0, freq_mhz=plan.freq_mhz, nchan=1024, total_s=0.14,
                       source="R3", ra="01:58:00.75", dec="65:43:00.3")
        parts.append(sigproc.read_fil(o.channelise(raws[i], cfg)))
        ocfgs.append(cfg)
    want = np.concatenate([p.data for p in parts], axis=2)
    assert got.data.shape == want.shape
    for col, (part, ocfg) in enumerate(zip(parts, ocfgs)):              # identical except at rounding ties (parity_util)
        pu.check_code_arrays(part.data, got.data[:, :, col * 1024:(col + 1) * 1024], ocfg)
    assert got.header["nchans"] == 2048 and got.header["fch1"] == pytest.approx(parts[0].header["fch1"])


def test_direct_scan_on_device_equals_spliced_per_if_runs(tmp_path):
    """frbch_run_scan (config-3 shape: several 32 MHz IFs, -d4, on ONE GPU): the IFall file written
    from the pitched device buffer is byte-identical to splice() of the per-IF files of the same library, and within
    the stated tolerance of the oracle."""
    d1, d2 = str(tmp_path / "direct"), str(tmp_path / "perif")
    os.makedirs(d1)
    os.makedirs(d2)
    raws, vd = {}, {}
    for i in (1, 2, 3, 4):
        raws[i] = synth.make_vdif(0.27, bw_mhz=32.0, nchan=1024, if_index=i)
        vd[i] = os.path.join(d1, f"x_ef_no0001_IF{i}.vdif")
        raws[i].tofile(vd[i])
    kw = dict(freq_lsb_0=1340.0, bw=32.0, nchan=1024, nsec=0.27, pol=4, tscrunch=2, source="R3", ra="01:58:00.75",
              dec="65:43:00.3")
    with contextlib.redirect_stdout(io.StringIO()):
        out = multi_if.process_scan(vd, out_dir=d1, direct=True, **kw)
        ref = multi_if.process_scan(vd, out_dir=d2, **kw)
    a, b = open(out, "rb").read(), open(ref, "rb").read()
    ga, gb = sigproc.read_fil(a), sigproc.read_fil(b)
    assert ga.header == gb.header
    assert ga.data.shape == gb.data.shape == (ga.data.shape[0], 4, 4096)
    assert np.array_equal(ga.data, gb.data)
    plan = multi_if.plan_ifs(4, 1340.0, 32.0)[3]           # highest IF: first 1024 columns
    cfg = o.Config(bw_mhz=32.0, freq_mhz=plan.freq_mhz, nchan=1024, total_s=0.27, pol_mode=4, tscrunch=2, source="R3",
                   ra="01:58:00.75", dec="65:43:00.3")
    want = sigproc.read_fil(o.channelise(raws[4], cfg)).data
    pu.check_code_arrays(want, ga.data[:, :, :1024], cfg)               # identical except at rounding ties


def test_node_scan_command_two_ranks_on_one_card(tmp_path):
    """python -m frb_baseband_amd.scan (base2fil.sh:30-67, 348-350, 404-448 in one command): two rank processes (rehearsal:
    both on GPU 0), 2 IFs each through frbch_run_scan into their FIFOs, the native join writes the IFall file; every IF's
    columns against the oracle.  Config-2 shape, 2 blocks per IF."""
    import subprocess
    import sys
    d = str(tmp_path)
    raws = {}
    for i in (1, 2, 3, 4):
        raws[i] = synth.make_vdif(0.14, bw_mhz=32.0, nchan=1024, if_index=i)
        raws[i].tofile(os.path.join(d, f"x_ef_no0001_IF{i}.vdif"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "frb_baseband_amd.scan", "--experiment", "x", "--st", "ef", "--scanname", "001", "--workdir", d,
           "--outdir", d, "--nif", "4", "--bw", "32", "--freqLSB_0", "1340.0", "--nchan", "1024", "--nsec", "0.14", "--source", "R3",
           "--ra", "01:58:00.75", "--dec", "65:43:00.3", "--gpus", "2", "--share-gpu"]
    pr = subprocess.run(cmd, cwd=root, env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr
    got = sigproc.read_fil(os.path.join(d, multi_if.ifall_name("x", "ef", "001", 2)))
    assert got.header["nchans"] == 4096
    for col, i in enumerate((4, 3, 2, 1)):
        plan = multi_if.plan_ifs(4, 1340.0, 32.0)[i - 1]
        cfg = o.Config(bw_mhz=32.0 if plan.sideband == "u" else -32.0, freq_mhz=plan.freq_mhz, nchan=1024, total_s=0.14,
                       source="R3", ra="01:58:00.75", dec="65:43:00.3")
        want = sigproc.read_fil(o.channelise(raws[i], cfg)).data
        pu.check_code_arrays(want, got.data[:, :, col * 1024:(col + 1) * 1024], cfg)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] / [3] at full length (10 s per IF) through size-independent properties (configurations 3 and
# 4 were oracle-checked at 2 blocks only)
# ------------------------------------------------------------------------------------------------------------------
def test_config3_full_length_scan_properties(hip_lib, ten_seconds):
    """10 s x IQUV x 3 IFs through frbch_scan_device (the bench's call, 152 blocks per IF):
    (1) every IF's columns of the scan's row buffer are bit-identical to that IF channelised alone (packed rows),
    (2) run-to-run bit identical, (3) the frozen-scale path (K2 digitises) reproduces the buffered first interval,
    (4) Stokes identity: every (time sample, channel) is ONE Jones vector, so I^2 = Q^2 + U^2 + V^2 to fp32 rounding."""
    # IFs 2 and 3: the same 10 s with the payloads of the frames rotated by 1000 / 2500 frames (other data in every block; the device
    # path takes frames as they are) -- generating two more 10-s streams cost 70 s of host time
    fr = ten_seconds.reshape(-1, 8032)
    raws = [ten_seconds]
    for shift in (1000, 2500):
        other = fr.copy()
        other[:, 32:] = np.roll(fr[:, 32:], shift, axis=0)
        raws.append(other.reshape(-1))
    bufs = [DeviceBuffer.from_numpy(r) for r in raws]
    nfr = raws[0].size // 8032
    chans = [ch.Channeliser(pu.lib_cfg(hip_lib, -32.0 if i % 2 else 32.0, 1024, 10.0, pol=5), hip_lib) for i in range(3)]
    info = chans[0].info
    nblocks = (nfr * 8000) // info.block_payload_bytes
    assert nblocks == 152
    rows = nblocks * info.rows_per_block
    out = DeviceBuffer(rows * 3 * info.row_bytes)
    scans = []
    for _ in range(2):
        for c in chans:
            c.reset()
        assert multi_if.scan_device(chans, [b.ptr.value for b in bufs], nfr, 8032, 32, 0, nblocks, out.ptr.value, rows) == rows
        scans.append(out.to_numpy(np.uint8).reshape(rows, 4, 3 * 1024))
    assert np.array_equal(scans[0], scans[1])                                  # (2)
    resc = [c.get_rescale() for c in chans]
    for c, (off, sc) in zip(chans, resc):
        c.reset()
        c.set_rescale(off, sc)
    assert multi_if.scan_device(chans, [b.ptr.value for b in bufs], nfr, 8032, 32, 0, nblocks, out.ptr.value, rows, flush=False) == rows
    assert np.array_equal(out.to_numpy(np.uint8).reshape(scans[0].shape), scans[0])   # (3)
    single = DeviceBuffer(rows * info.row_bytes)
    for i, c in enumerate(chans):                                              # (1)
        c.reset()
        r1 = c.process_device(bufs[i].ptr.value, nfr, 8032, 32, 0, nblocks, single.ptr.value, single.nbytes)
        r1 += c.flush_device(single.ptr.value + r1 * info.row_bytes, single.nbytes - r1 * info.row_bytes)
        assert r1 == rows
        assert np.array_equal(single.to_numpy(np.uint8).reshape(rows, 4, 1024), scans[0][:, :, i * 1024:(i + 1) * 1024])
    d = scans[0].astype(np.float64)
    assert abs(d[:, 0].mean() - 127.5) < 0.6 and abs(d[:, 0].std() - 127.5 / 6) < 1.5       # I: mean / sigma of the digitiser
    # (4) on the float products of 19 blocks
    c = chans[0]
    c.reset()
    nb = 19
    pw = DeviceBuffer(nb * info.rows_per_block * 4 * 1024 * 4)
    c.power_device(bufs[0].ptr.value, nfr, 8032, 32, 40 * info.block_payload_bytes, nb, pw.ptr.value, pw.nbytes)
    p = pw.to_numpy(np.float32).reshape(-1, 4, 1024).astype(np.float64)
    lhs, rhs = p[:, 0] ** 2, p[:, 1] ** 2 + p[:, 2] ** 2 + p[:, 3] ** 2
    assert np.abs(lhs - rhs).max() <= 4e-6 * lhs.max()
    assert np.all(p[:, 0] >= 0)
    for c in chans:
        c.close()


def test_config4_full_length_parseval_and_determinism(hip_lib):
    """BASELINE configs[3], one IF of a GPU's share at full length: 10 s x 64 MHz -> 4096 channels (19 blocks of 2^26
    samples): Parseval per block on the float power (as test_parseval_every_block_full_size), `-t 8` codes run-to-run
    identical and equal between the buffered first interval and the frozen-scale path."""
    bw, nchan, r = 64.0, 4096, 8192
    # 10 s = 19 blocks of 2^26 samples: two generated seconds, repeated (the blocks straddle the seams, so no two are alike; the
    # properties below hold for any bytes, and generating 10 s of a 64 MHz IF cost a minute of host time)
    two = synth.make_vdif(2.0, bw_mhz=bw, nchan=nchan)
    raw = np.tile(two, 5)
    n = 2 * nchan * r
    payload = o.strip_frames(raw, 8032, 32)
    nblocks = payload.size * 2 // n
    assert nblocks == 19
    d_raw = DeviceBuffer.from_numpy(raw)
    nfr = raw.size // 8032
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, 10.0), hip_lib) as c:
        info = c.info
        pw = DeviceBuffer(nblocks * info.rows_per_block * nchan * 4)
        c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
        got = pw.to_numpy(np.float32).reshape(nblocks, -1).astype(np.float64).sum(axis=1)
    want = np.empty(nblocks)
    for b in range(nblocks):
        x = o.unpack_2bit(payload[b * n // 2:(b + 1) * n // 2])
        e = 0.0
        for pol in range(2):
            xs = x[pol]
            alt = xs[0::2].sum() - xs[1::2].sum()
            e += (n * (xs * xs).sum() + xs.sum() ** 2 - alt ** 2) / 2.0
        want[b] = r * e
    np.testing.assert_allclose(got, want, rtol=2e-6)
    del pw
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, 10.0, tscr=8), hip_lib) as c:
        info = c.info
        rows = nblocks * info.rows_per_block
        out = DeviceBuffer(rows * info.row_bytes)
        outs = []
        for _ in range(2):
            c.reset()
            r1 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
            r1 += c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
            assert r1 == rows
            outs.append(out.to_numpy(np.uint8))
        off, sc = c.get_rescale()
        c.reset()
        c.set_rescale(off, sc)
        assert c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes) == rows
        fused = out.to_numpy(np.uint8)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], fused)
