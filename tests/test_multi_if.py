"""IF sharding over ranks + host-side frequency concatenation (replaces splice, base2fil.sh:422).
world_size-2 gloo run on CPU; the per-IF compute goes through the TEST-ONLY emulator library."""
import os
import subprocess
import sys

import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import multi_if, sigproc, synth
from tests import parity_util as pu
from oracle import frb_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frequency_plan_matches_base2fil():
    # base2fil.sh:54,65,254: LSB series from freqLSB_0 step 2bw; USB series from freqLSB_0+bw
    plans = multi_if.plan_ifs(4, 1340.49, 32.0)
    assert [(p.index, p.sideband) for p in plans] == [(1, "l"), (2, "u"), (3, "l"), (4, "u")]
    assert [p.freq_mhz for p in plans] == pytest.approx([1340.49, 1372.49, 1404.49, 1436.49])
    assert multi_if.splice_order(4) == [4, 3, 2, 1]
    assert multi_if.shard(16, 8, 0) == [1, 9] and multi_if.shard(16, 8, 7) == [8, 16]
    assert sorted(sum((multi_if.shard(5, 2, r) for r in range(2)), [])) == [1, 2, 3, 4, 5]
    assert multi_if.ifall_name("pr001a", "ef", "001", 2) == "pr001a_ef_no0001_IFall_vdif_pol2.fil"


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from frb_baseband_amd import multi_if, _lib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = _lib.load(os.path.join(sys.argv[1], "tests", "emu", "libfrbch_emu.so"))
d = sys.argv[2]
vd = {i: os.path.join(d, f"x_ef_no0001_IF{i}.vdif") for i in (1, 2, 3, 4)}
out = multi_if.process_scan(vd, freq_lsb_0=1340.0, bw=16.0, nchan=32, nsec=0.02, out_dir=d, rank=rank, world=world,
                            barrier=dist.barrier, lib=lib, ra="01:00:00.0", dec="02:00:00.0")
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_scan_equals_oracle_splice(emu_lib, tmp_path):
    d = str(tmp_path)
    raws = {}
    for i in (1, 2, 3, 4):
        raws[i] = synth.make_vdif(0.02, bw_mhz=16.0, nchan=32, if_index=i)
        raws[i].tofile(os.path.join(d, f"x_ef_no0001_IF{i}.vdif"))
    script = os.path.join(d, "worker.py")
    open(script, "w").write(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, script, ROOT, d], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = sigproc.read_fil(os.path.join(d, "IFall.fil"))
    # oracle: channelise every IF with the base2fil frequency plan, concatenate highest IF first
    parts = []
    for i in (4, 3, 2, 1):
        plan = multi_if.plan_ifs(4, 1340.0, 16.0)[i - 1]
        bw = 16.0 if plan.sideband == "u" else -16.0
        cfg = o.Config(bw_mhz=bw, freq_mhz=plan.freq_mhz, nchan=32, total_s=0.02, source="unknown")
        parts.append(sigproc.read_fil(o.channelise(raws[i], cfg)))
    want = np.concatenate([p.data for p in parts], axis=2)
    assert got.data.shape == want.shape == (parts[0].data.shape[0], 1, 128)
    assert np.count_nonzero(got.data != want) <= 2e-5 * want.size      # +-1 at rounding ties only
    assert np.abs(got.data.astype(int) - want.astype(int)).max() <= 1
    assert got.header["nchans"] == 128
    assert got.header["fch1"] == pytest.approx(parts[0].header["fch1"])
    fch = [p.header["fch1"] for p in parts]
    assert fch == sorted(fch, reverse=True)                 # descending frequency across IFs
    assert fch[0] - fch[1] == pytest.approx(16.0)


def _oracle_ifall(raws, nif, bw, nchan, secs, **kw):
    parts = []
    for i in range(nif, 0, -1):
        plan = multi_if.plan_ifs(nif, 1340.0, bw)[i - 1]
        cfg = o.Config(bw_mhz=bw if plan.sideband == "u" else -bw, freq_mhz=plan.freq_mhz, nchan=nchan, total_s=secs,
                       source="unknown", **kw)
        parts.append(sigproc.read_fil(o.channelise(raws[i], cfg)))
    return parts, np.concatenate([p.data for p in parts], axis=2)


def _direct_scan(lib, tmp_path, nif, bw, nchan, secs, pol=2, tscrunch=1, nbit=8, lengths=None):
    d = str(tmp_path)
    raws, vd = {}, {}
    for i in range(1, nif + 1):
        raws[i] = synth.make_vdif((lengths or {}).get(i, secs), bw_mhz=bw, nchan=nchan, if_index=i)
        vd[i] = os.path.join(d, f"x_ef_no0001_IF{i}.vdif")
        raws[i].tofile(vd[i])
    out = multi_if.process_scan(vd, freq_lsb_0=1340.0, bw=bw, nchan=nchan, nsec=secs, out_dir=d, lib=lib, direct=True,
                                pol=pol, tscrunch=tscrunch, nbit=nbit, ra="01:00:00.0", dec="02:00:00.0")
    assert sorted(f for f in os.listdir(d) if f.endswith(".fil")) == ["IFall.fil"]      # no per-IF products
    return raws, sigproc.read_fil(out)


def test_direct_scan_on_one_device_equals_oracle_splice(emu_lib, tmp_path):
    """frbch_run_scan (8f row 1): 4 IFs concatenated in device memory == splice of the per-IF oracle outputs."""
    raws, got = _direct_scan(emu_lib, tmp_path, 4, 16.0, 32, 0.02)
    parts, want = _oracle_ifall(raws, 4, 16.0, 32, 0.02)
    assert got.data.shape == want.shape == (parts[0].data.shape[0], 1, 128)
    assert np.abs(got.data.astype(int) - want.astype(int)).max() <= 1
    assert np.count_nonzero(got.data != want) <= 2e-5 * want.size
    assert got.header["nchans"] == 128 and got.header["nifs"] == 1
    assert got.header["fch1"] == pytest.approx(parts[0].header["fch1"])
    assert got.header["tstart"] == pytest.approx(parts[0].header["tstart"], abs=1e-12)


def test_direct_scan_four_products_and_unequal_inputs(emu_lib, tmp_path):
    """-d4 -t2 16-bit rows ([t][product][IF chans]); the second IF is one block longer: rows cut to the shortest."""
    raws, got = _direct_scan(emu_lib, tmp_path, 3, 16.0, 32, 0.02, pol=4, tscrunch=2, nbit=16, lengths={2: 0.03})
    parts, want = _oracle_ifall(raws, 3, 16.0, 32, 0.02, pol_mode=4, tscrunch=2, nbit=16)
    assert got.data.shape == want.shape == (parts[0].data.shape[0], 4, 96)
    assert np.abs(got.data.astype(int) - want.astype(int)).max() <= 1
    # 16-bit codes: +-1 at rounding ties within the stated tolerance (parity_util: 1e-5 of the samples per code unit of sigma)
    assert np.count_nonzero(got.data != want) <= 1.0e-5 * (32767.5 / 6.0) * want.size


def test_direct_scan_rejects_mixed_geometry(emu_lib, tmp_path):
    from frb_baseband_amd import channeliser as ch
    a = ch.Channeliser(ch.new_config(emu_lib, bw_mhz=16.0, nchan=32, total_s=0.02), emu_lib)
    b = ch.Channeliser(ch.new_config(emu_lib, bw_mhz=16.0, nchan=64, total_s=0.02), emu_lib)
    with pytest.raises(ch.InputError, match="share"):
        multi_if.run_scan([a, b], ["/nonexistent/a.vdif", "/nonexistent/b.vdif"], str(tmp_path / "o.fil"))
    a.close()
    b.close()


@pytest.mark.parametrize("kw", [
    dict(nchan=32, freq_res=64),                                   # interval = scan: everything digitised in the flush
    dict(nchan=32, freq_res=64, pol=5, interval=0.004, maxb=3),    # four products, interval inside the scan, small batches
    dict(nchan=32, freq_res=64, pol=4, nbit=2, tscr=2, interval=0.0),
    dict(nchan=32, freq_res=64, nbit=16, interval=0.004, const=0, maxb=2),
    dict(nchan=16, dm=1.0, coherent=1, freq=316.0, pol=4, tscr=2, maxb=2),   # the coherent filterbank (K4 writes the rows) into the scan's columns
    dict(nchan=128, freq_res=512, pol=5, overlap=176 | (3 << 24)),         # the digitiser beside the next IF's K1 on plain streams (mode 3: the automatic setting's chain; emulator: queue order)
])
def test_scan_device_rows_are_the_splice_of_the_per_if_rows(emu_lib, kw):
    """frbch_scan_device: every IF's rows land in its columns of one row buffer == the per-IF outputs side by side."""
    kw = dict(kw)
    overlap = kw.pop("overlap", 0)
    nchan, bw, secs, nif_scan = kw.pop("nchan"), 16.0, 0.03, 3
    raws = [synth.make_vdif(secs, bw_mhz=bw, nchan=nchan, if_index=i) for i in range(nif_scan)]
    chans = []
    for i in range(nif_scan):
        cfg = pu.lib_cfg(emu_lib, -bw if i % 2 else bw, nchan, secs, **kw)
        cfg.overlap = overlap
        chans.append(ch.Channeliser(cfg, emu_lib))
    info = chans[0].info
    nfr = raws[0].size // 8032
    nblocks = (nfr * 8000 - info.block_payload_bytes) // info.block_stride_bytes + 1     # (overlap-save when coherent)
    rows = nblocks * info.rows_per_block
    # per IF, packed rows
    single = []
    for c, raw in zip(chans, raws):
        out = np.zeros(rows * info.row_bytes, np.uint8)
        r1 = c.process_device(raw.ctypes.data, nfr, 8032, 32, 0, nblocks, out.ctypes.data, out.size)
        r1 += c.flush_device(out.ctypes.data + r1 * info.row_bytes, out.size - r1 * info.row_bytes)
        assert r1 == rows
        single.append(out.reshape(rows, info.nif, -1))
        c.reset()
    want = np.concatenate(single, axis=2)
    # the scan: one row buffer, in two calls (the second one flushes)
    buf = np.zeros(rows * nif_scan * info.row_bytes, np.uint8)
    half = nblocks // 2
    ptrs = [r.ctypes.data for r in raws]
    got1 = multi_if.scan_device(chans, ptrs, nfr, 8032, 32, 0, half, buf.ctypes.data, rows, flush=False)
    off = got1 * nif_scan * info.row_bytes
    got2 = multi_if.scan_device(chans, ptrs, nfr, 8032, 32, half * info.block_stride_bytes, nblocks - half,
                                buf.ctypes.data + off, rows - got1, flush=True)
    assert got1 + got2 == rows
    np.testing.assert_array_equal(buf.reshape(want.shape), want)
    with pytest.raises(ch.InputError):
        multi_if.scan_device(chans, ptrs, nfr, 8032, 32, 0, nblocks + 1, buf.ctypes.data, rows)
    for c in chans:
        c.close()


def _scan_null_stream_last_handle_first(lib, nchan=128, freq_res=512, pol=5, secs=0.05, nif_scan=3):
    """A scan queued with stream == NULL runs on ifs[0]'s stream; the LAST handle is then read and reset first.  Its own stream
    is idle, so without the scan's event recorded on it (frbch_scan_device marks every handle whose stream is not the scan's)
    the reset would overwrite offset / scale while the scan's statistics and digitiser still run."""
    raws = [synth.make_vdif(secs, bw_mhz=16.0, nchan=nchan, if_index=i) for i in range(nif_scan)]
    chans = [ch.Channeliser(pu.lib_cfg(lib, 16.0, nchan, secs, pol=pol, freq_res=freq_res), lib) for _ in range(nif_scan)]
    info = chans[0].info
    nfr = raws[0].size // 8032
    nblocks = (nfr * 8000 - info.block_payload_bytes) // info.block_stride_bytes + 1
    rows = nblocks * info.rows_per_block
    return raws, chans, info, nfr, nblocks, rows


def test_scan_device_null_stream_then_last_handle_first(emu_lib):
    raws, chans, info, nfr, nblocks, rows = _scan_null_stream_last_handle_first(emu_lib)
    ptrs = [r.ctypes.data for r in raws]
    bufs, scales = [], []
    for _ in range(2):
        buf = np.zeros(rows * len(chans) * info.row_bytes, np.uint8)
        assert multi_if.scan_device(chans, ptrs, nfr, 8032, 32, 0, nblocks, buf.ctypes.data, rows, flush=True, stream=None) == rows
        scales.append(chans[-1].get_rescale())          # the last handle first
        for c in reversed(chans):
            c.reset()
        bufs.append(buf)
    np.testing.assert_array_equal(bufs[0], bufs[1])
    np.testing.assert_array_equal(scales[0][1], scales[1][1])
    assert bufs[0].any()
    for c in chans:
        c.close()


# ------------------------------------------------------------------------------------------------------------------
# the node-level scan: python -m frb_baseband_amd.scan = one rank process per GPU + the native streaming join
# (base2fil.sh:30-67, 348-350, 404-448).  On CPU the ranks load the TEST-ONLY emulator build through FRBCH_LIB.
# ------------------------------------------------------------------------------------------------------------------
def _node_scan(tmp_path, emu_lib, nif, world, extra=(), drop=None, **kw):
    from frb_baseband_amd import _lib
    d = str(tmp_path)
    raws = {}
    for i in range(1, nif + 1):
        raws[i] = synth.make_vdif(0.02, bw_mhz=16.0, nchan=32, if_index=i)
        if i != drop:
            raws[i].tofile(os.path.join(d, f"x_ef_no0001_IF{i}.vdif"))
    emu = os.path.join(ROOT, "tests", "emu", "libfrbch_emu.so")
    env = dict(os.environ, FRBCH_LIB=emu, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "frb_baseband_amd.scan", "--experiment", "x", "--st", "ef", "--scanname", "001", "--workdir", d,
           "--outdir", d, "--nif", str(nif), "--bw", "16", "--freqLSB_0", "1340.0", "--nchan", "32", "--nsec", "0.02",
           "--ra", "01:00:00.0", "--dec", "02:00:00.0", "--gpus", str(world), "--share-gpu"] + list(extra)
    pr = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    return raws, pr, os.path.join(d, multi_if.ifall_name("x", "ef", "001", kw.get("pol", 2)))


def test_node_scan_command_two_ranks_and_the_native_join(emu_lib, tmp_path):
    raws, pr, out = _node_scan(tmp_path, emu_lib, 4, 2)
    assert pr.returncode == 0, pr.stderr
    got = sigproc.read_fil(out)
    parts, want = _oracle_ifall(raws, 4, 16.0, 32, 0.02)
    assert got.data.shape == want.shape == (parts[0].data.shape[0], 1, 128)
    assert np.abs(got.data.astype(int) - want.astype(int)).max() <= 1 and np.count_nonzero(got.data != want) <= 2e-5 * want.size
    assert got.header["nchans"] == 128 and got.header["fch1"] == pytest.approx(parts[0].header["fch1"])
    assert not [f for f in os.listdir(str(tmp_path)) if f.endswith(".fil") and "IFall" not in f]   # no per-IF products
    # the same scan on one rank: the rank writes the product itself, byte for byte the same file
    one = open(out, "rb").read()
    os.remove(out)
    _r, pr1, out1 = _node_scan(tmp_path, emu_lib, 4, 1)
    assert pr1.returncode == 0, pr1.stderr
    assert open(out1, "rb").read() == one


def test_node_scan_three_ranks_four_products_uneven_shares(emu_lib, tmp_path):
    raws, pr, out = _node_scan(tmp_path, emu_lib, 5, 3, extra=("--pol", "5", "--nbit", "16", "--tscrunch", "2"), pol=5)
    assert pr.returncode == 0, pr.stderr
    got = sigproc.read_fil(out)
    parts, want = _oracle_ifall(raws, 5, 16.0, 32, 0.02, pol_mode=5, nbit=16, tscrunch=2)
    assert got.data.shape == want.shape and got.header["nifs"] == 4 and got.header["nchans"] == 160
    _m, dscale, _v = o.digi_params(16)                                  # 16-bit codes: rounding ties are that much denser (parity_util)
    assert np.abs(got.data.astype(int) - want.astype(int)).max() <= 1
    assert np.count_nonzero(got.data != want) <= pu.MISMATCH_FRAC_PER_SIGMA * dscale * want.size


def test_node_scan_a_failing_rank_fails_the_run(emu_lib, tmp_path):
    _r, pr, out = _node_scan(tmp_path, emu_lib, 4, 2, drop=1)       # IF1's VDIF is missing: rank 1 fails
    assert pr.returncode != 0
    assert "rank" in pr.stderr and "failed" in pr.stderr


def test_native_join_cuts_to_the_shortest_piece_and_checks_headers(emu_lib, tmp_path):
    """frbch_join == multi_if.splice (the numpy restatement of sigproc splice) on files of unequal length; pieces that do not
    continue each other in frequency (wrong order, a gap) are refused instead of being mislabelled"""
    d = str(tmp_path)
    pieces = []
    # three USB pieces of 16 MHz, highest first (base2fil.sh:350,367): centres 1640, 1624, 1608 MHz
    for i, (secs, freq) in enumerate(((0.03, 1640.0), (0.02, 1624.0), (0.03, 1608.0)), start=1):
        raw = synth.make_vdif(secs, bw_mhz=16.0, nchan=32, if_index=i)
        with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, secs, pol=4, freq_res=64, freq=freq), emu_lib) as c:
            p = os.path.join(d, f"p{i}.fil")
            open(p, "wb").write(c.channelise_bytes(raw))
            pieces.append(p)
    join = os.path.join(ROOT, "frb_baseband_amd", "csrc", "frbch_join")
    out = os.path.join(d, "joined.fil")
    pr = subprocess.run([join, out] + pieces, capture_output=True, text=True, timeout=120)
    assert pr.returncode == 0, pr.stderr
    assert open(out, "rb").read() == multi_if.splice(pieces)
    got = sigproc.read_fil(out)
    assert got.header["nchans"] == 96 and got.header["fch1"] == pytest.approx(sigproc.read_fil(pieces[0]).header["fch1"])
    # wrong order / a missing piece: the header of piece 0 would label the channels wrongly
    pr = subprocess.run([join, out, pieces[1], pieces[0], pieces[2]], capture_output=True, text=True, timeout=120)
    assert pr.returncode != 0 and "does not continue" in pr.stderr
    pr = subprocess.run([join, out, pieces[0], pieces[2]], capture_output=True, text=True, timeout=120)
    assert pr.returncode != 0 and "does not continue" in pr.stderr
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.02, pol=2, freq_res=64, freq=1624.0), emu_lib) as c:
        other = os.path.join(d, "other.fil")
        open(other, "wb").write(c.channelise_bytes(synth.make_vdif(0.02, bw_mhz=16.0, nchan=32)))
    pr = subprocess.run([join, out, pieces[0], other], capture_output=True, text=True, timeout=120)
    assert pr.returncode != 0 and "differ" in pr.stderr
