"""CPU tests of the C-ABI host logic and of the generic kernels' index arithmetic, run through the
TEST-ONLY host emulator build (tests/emu): same engine source, kernels executed by loops.
The GPU parity tests proper are in test_gpu_parity.py."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch
from frb_baseband_amd import sigproc, synth
from oracle import frb_oracle as o
from tests import parity_util as pu

CASES = [
    (16.0, 128, 0.03, {}),
    (-16.0, 128, 0.03, {}),
    (16.0, 64, 0.03, dict(pol=4)),
    (-16.0, 32, 0.03, dict(pol=4, nbit=-32, tscr=4, freq_res=64)),
    (16.0, 32, 0.03, dict(pol=0, nbit=2, tscr=2, freq_res=64)),
    (16.0, 32, 0.03, dict(pol=3, nbit=16, freq_res=64, interval=0.0)),
    (16.0, 32, 0.03, dict(pol=1, freq_res=64, interval=0.004, const=0)),
    (16.0, 32, 0.03, dict(freq_res=64, interval=0.004, const=1, maxb=3)),
    (16.0, 16, 0.02, dict(freq_res=16, tscr=16)),            # tscrunch == freq_res: one row per block
    (32.0, 256, 0.03, dict(tscr=8)),                          # tscrunch > K2 sub-tile: LDS accumulators
    (16.0, 32, 0.03, dict(freq_res=64, start=0.01)),          # -S inside the file
    (16.0, 32, 0.03, dict(freq_res=64, bits=1)),              # 1-bit mode VDIF_8000-1024-16-1 (spif2file.sh:58-61)
    (-16.0, 64, 0.03, dict(pol=4, bits=1, start=0.01 + 2 / 32e6)),  # 1 bit, LSB, -S not on a byte boundary
    (16.0, 32, 0.03, dict(freq_res=64, payload_bytes=10000)), # Mark5B-sized payload (spif2file.sh mode table)
    (16.0, 32, 0.03, dict(freq_res=64, payload_bytes=1000, legacy=1, bits=1)),
    # coherent dedispersion -D <dm> -F C:D (process_vdif.py:177-180): overlap-save blocks, K1 -> K2c -> K3 -> K4
    (16.0, 16, 0.012, dict(dm=1.0, coherent=1, freq=316.0)),
    (-16.0, 16, 0.012, dict(dm=1.0, coherent=1, freq=316.0, pol=4, tscr=2, maxb=2)),
    (16.0, 16, 0.012, dict(dm=1.0, coherent=1, freq=316.0, pol=3, nbit=2, tscr=8, interval=0.004, const=0)),
    (-16.0, 32, 0.012, dict(dm=0.7, coherent=1, freq=330.0, nbit=-32, start=0.001, bits=1)),
    (16.0, 64, 0.02, dict(dm=56.7, coherent=1, freq=1400.0, nbit=16, freq_res=128)),   # tiny smearing: 1 + 1 samples
    # dynamic level setting (cfg.unpack_mode 1, the optional Jenet-Anderson unpack: DESIGN.md section 2a): windows of 512 and of 64
    # samples, excision on (default 10 sigma), tight (2 sigma: many windows zeroed) and off, another threshold, frames of 1000 bytes
    # (windows straddle frame boundaries), -S inside the file, four products, coherent
    (16.0, 32, 0.03, dict(freq_res=64, dynamic={})),
    (-16.0, 64, 0.03, dict(pol=4, dynamic=dict(nsample=64, cutoff_sigma=2.0))),
    (16.0, 32, 0.03, dict(freq_res=64, nbit=-32, start=0.01, payload_bytes=1000, legacy=1, dynamic=dict(cutoff_sigma=-1.0, threshold=1.2))),
    (16.0, 16, 0.012, dict(dm=1.0, coherent=1, freq=316.0, maxb=2, dynamic=dict(nsample=32))),
]


@pytest.mark.parametrize("bw,nchan,secs,kw", CASES)
def test_fil_matches_oracle(emu_lib, bw, nchan, secs, kw):
    pu.run_streaming_case(emu_lib, bw, nchan, secs, **kw)


def test_ragged_pushes_equal_single_push(emu_lib):
    raw = synth.make_vdif(0.03, bw_mhz=16.0, nchan=32)
    cfg = pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64, interval=0.004, const=1)
    with ch.Channeliser(cfg, emu_lib) as c:
        whole = c.channelise_bytes(raw)
    rng = np.random.default_rng(5)
    with ch.Channeliser(cfg, emu_lib) as c:
        pos, body = 0, bytearray()
        while pos < raw.size:
            n = int(rng.integers(1, 30000))
            c.push(raw[pos:pos + n])
            pos += n
            body += c.pull()
        c.flush()
        body += c.pull()
        ragged = c.sigproc_header() + bytes(body)
    assert ragged == whole


def test_total_seconds_limits_output(emu_lib):
    raw = synth.make_vdif(0.03, bw_mhz=16.0, nchan=32)
    for secs in (0.004, 0.0101):
        ocfg = pu.oracle_cfg(16.0, 32, secs, freq_res=64)
        ref = o.channelise(raw, ocfg)
        with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, secs, freq_res=64), emu_lib) as c:
            got = c.channelise_bytes(raw)
        pu.check_codes(ref, got, ocfg)


def test_input_bits_pinned_or_automatic(emu_lib):
    raw1 = synth.make_vdif(0.03, bw_mhz=16.0, nchan=32, bits=1)
    auto = pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64)
    assert auto.input_bits == 0
    with ch.Channeliser(auto, emu_lib) as c:
        want = c.channelise_bytes(raw1)
        assert c.get_info().block_payload_bytes == 2 * 32 * 64 // 4          # N/4 bytes per block at 1 bit
    pinned = pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64)
    pinned.input_bits = 1
    with ch.Channeliser(pinned, emu_lib) as c:
        assert c.channelise_bytes(raw1) == want
    pinned.input_bits = 2                                                    # stream says 1: refuse, don't misread
    with ch.Channeliser(pinned, emu_lib) as c:
        with pytest.raises(ch.RunError):
            c.channelise_bytes(raw1)
    pinned.input_bits = 3
    with pytest.raises(ch.InputError):
        ch.Channeliser(pinned, emu_lib)


def test_empty_and_short_inputs(emu_lib):
    cfg = pu.lib_cfg(emu_lib, 16.0, 32, 1.0, freq_res=64)
    with ch.Channeliser(cfg, emu_lib) as c:                  # nothing pushed
        c.flush()
        assert c.pull() == b""
        with pytest.raises(ch.RunError):
            c.sigproc_header()
    raw = synth.make_vdif(1.0 / 16000, bw_mhz=16.0, nchan=32, payload_bytes=1000)   # 1 frame < one block
    with ch.Channeliser(cfg, emu_lib) as c:
        fil = c.channelise_bytes(raw)
    f = sigproc.read_fil(fil)
    assert f.data.shape[0] == 0 and f.header["nchans"] == 32


def test_legacy_headers_and_other_payload(emu_lib):
    for legacy, payload in ((1, 8000), (0, 1000)):
        raw = synth.make_vdif(0.02, bw_mhz=16.0, nchan=32, legacy=legacy, payload_bytes=payload)
        ocfg = pu.oracle_cfg(16.0, 32, 0.02, freq_res=64)
        ref = o.channelise(raw, ocfg)
        with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.02, freq_res=64), emu_lib) as c:
            got = c.channelise_bytes(raw)
            assert c.get_info().header_bytes == (16 if legacy else 32)
        pu.check_codes(ref, got, ocfg)


def test_rejects_unsupported_streams(emu_lib):
    raw = synth.make_vdif(0.002, bw_mhz=16.0, nchan=32).copy()
    raw[15] = (raw[15] & 0x83) | (7 << 2)                    # bits/sample-1 = 7 -> 8-bit samples
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 1.0, freq_res=64), emu_lib) as c:
        with pytest.raises(ch.RunError):
            c.push(raw)
    for bad in (dict(nchan=100), dict(tscrunch=3), dict(nbit_out=4), dict(pol_mode=7),
                dict(coherent=1, dm=5000.0, freq_mhz=300.0, nchan=16),      # smearing beyond the largest freq_res
               
                dict(bw_mhz=0.0), dict(nchan=32, freq_res=64, tscrunch=128)):
        with pytest.raises(ch.InputError):
            ch.Channeliser(ch.new_config(emu_lib, **bad), emu_lib)


def test_set_rescale_gives_reproducible_codes(emu_lib):
    """given the same scale the digitised integers are reproducible (SURVEY 7 hard part 6)."""
    raw = synth.make_vdif(0.03, bw_mhz=16.0, nchan=32)
    cfg = pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64)
    with ch.Channeliser(cfg, emu_lib) as c:
        a = c.channelise_bytes(raw)
        off, sc = c.get_rescale()
    with ch.Channeliser(cfg, emu_lib) as c:
        c.set_rescale(off, sc)                               # fused path from the first block on
        b = c.channelise_bytes(raw)
    assert a == b
    ocfg = pu.oracle_cfg(16.0, 32, 0.03, freq_res=64)
    ocfg.fixed_offset, ocfg.fixed_scale = off, sc
    pu.check_codes(o.channelise(raw, ocfg), b, ocfg)


def test_run_file_into_fifo(emu_lib, tmp_path):
    """-o may be a pre-made FIFO (base2fil.sh:348-349): opened without O_EXCL, written sequentially,
    never unlinked (INSTALL.md:32-35; process_vdif.py:146-149)."""
    raw = synth.make_vdif(0.03, bw_mhz=16.0, nchan=32)
    vd = tmp_path / "a_IF1.vdif"
    raw.tofile(vd)
    fifo = str(tmp_path / "a_IF1.vdif_pol2.fil")
    os.mkfifo(fifo)
    got = {}

    def reader():
        with open(fifo, "rb") as f:
            got["data"] = f.read()
    th = threading.Thread(target=reader)
    th.start()
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64), emu_lib) as c:
        c.run_file(str(vd), fifo)
    th.join(timeout=30)
    assert os.path.exists(fifo)
    ocfg = pu.oracle_cfg(16.0, 32, 0.03, freq_res=64)
    pu.check_codes(o.channelise(raw, ocfg), got["data"], ocfg)
    # regular file target: truncated and rewritten
    out = str(tmp_path / "o.fil")
    open(out, "wb").write(b"x" * 10_000_000)
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64), emu_lib) as c:
        c.run_file(str(vd), out)
    assert open(out, "rb").read() == got["data"]
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.03, freq_res=64), emu_lib) as c:
        with pytest.raises(ch.RunError):
            c.run_file(str(tmp_path / "missing.vdif"), out)


def test_fifo_takes_large_outputs_by_reference(emu_lib, tmp_path, monkeypatch):
    """a FIFO sink (base2fil.sh:348-350) with more than two pipes' worth per slot: the rows go out through vmsplice (frbch_info.diag
    bit 1), the last pipe's worth of every slot by write(); same bytes as the regular-file output, and as with FRBCH_FIFO_COPY=1"""
    raw = synth.make_vdif(0.04, bw_mhz=16.0, nchan=32)
    vd = tmp_path / "a_IF1.vdif"
    raw.tofile(vd)
    kw = dict(freq_res=64, pol=4, nbit=-32)          # 512-byte rows: ~10 MB of output
    out = str(tmp_path / "o.fil")
    with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.04, **kw), emu_lib) as c:
        c.run_file(str(vd), out)
    want = open(out, "rb").read()
    assert len(want) > 4 << 20
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
        with ch.Channeliser(pu.lib_cfg(emu_lib, 16.0, 32, 0.04, **kw), emu_lib) as c:
            c.run_file(str(vd), fifo)
            diag = c.get_info().diag
        th.join(timeout=60)
        assert got["data"] == want
        assert bool(diag & 2) == (not copy)


def test_device_entry_points_and_power_tap(emu_lib):
    """frbch_process_device / flush_device / power_device semantics (emulator: host pointers)."""
    bw, nchan, r = 16.0, 32, 64
    raw = synth.make_vdif(0.02, bw_mhz=bw, nchan=nchan)
    with ch.Channeliser(pu.lib_cfg(emu_lib, bw, nchan, 0.02, freq_res=r), emu_lib) as c:
        info = c.info
        nfr = raw.size // 8032
        nblocks = (nfr * 8000) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        out = np.zeros(rows * info.row_bytes, np.uint8)
        r1 = c.process_device(raw.ctypes.data, nfr, 8032, 32, 0, nblocks, out.ctypes.data, out.size)
        assert r1 == 0                                       # interval still being measured
        r2 = c.flush_device(out.ctypes.data, out.size)
        assert r2 == rows
        with pytest.raises(ch.InputError):
            c.process_device(raw.ctypes.data, nfr, 8032, 32, 0, nblocks + 1, out.ctypes.data, out.size)
        c.reset()
        pw = np.zeros(rows * info.nif * nchan, np.float32)
        c.power_device(raw.ctypes.data, nfr, 8032, 32, 0, nblocks, pw.ctypes.data, pw.nbytes)
    ocfg = pu.oracle_cfg(bw, nchan, 0.02, freq_res=r)
    ref = sigproc.read_fil(o.channelise(raw, ocfg))
    assert np.count_nonzero(out.reshape(ref.data.shape) != ref.data) <= 2
    want = ocfg.result["power"][:, ::-1, :].transpose(2, 0, 1).reshape(-1)   # USB: flipped
    scale = ocfg.result["power"].mean()
    assert np.abs(pw - want).max() <= pu.POWER_RTOL * scale


def test_frame_header_checks(emu_lib):
    """invalid frames read as zero voltages, missing frame numbers are filled with zero frames (the stream stays
    contiguous in time: -cont, process_vdif.py:157), both are counted; a header whose geometry changes mid-stream is an
    error."""
    raw = synth.make_vdif(0.012, bw_mhz=16.0, nchan=32).copy()
    cfg = pu.lib_cfg(emu_lib, 16.0, 32, 0.012, freq_res=64)
    with ch.Channeliser(cfg, emu_lib) as c:
        clean = c.channelise_bytes(raw)
        assert c.get_info().frames_seen == raw.size // 8032 and c.get_info().frame_gaps == 0
    marked = raw.copy()
    marked[3 * 8032 + 3] |= 0x80                                  # invalid bit of frame 3
    marked = np.concatenate([marked[: 7 * 8032], marked[9 * 8032:]])   # frames 7 and 8 are missing
    ocfg = pu.oracle_cfg(16.0, 32, 0.012, freq_res=64)
    ref = o.channelise(marked, ocfg)
    assert ocfg.result["frame_counters"] == dict(gaps=1, filled=2, invalid=1)
    with ch.Channeliser(cfg, emu_lib) as c:
        got = c.channelise_bytes(marked)
        info = c.get_info()
    assert info.frames_invalid == 1 and info.frame_gaps == 1 and info.frames_filled == 2
    assert len(got) == len(clean)                                  # same number of samples: the gap was filled
    pu.check_codes(ref, got, ocfg)
    fg, fc = sigproc.read_fil(got), sigproc.read_fil(clean)
    assert np.count_nonzero(fg.data != fc.data) > 1000             # ... and the zeroed frames do change the output
    # the same pushed in ragged pieces (a gap may fall on a push boundary)
    with ch.Channeliser(cfg, emu_lib) as c:
        pos, body = 0, bytearray()
        for n in (5000, 8032 * 7 - 5000 + 11, 9000, marked.size):
            c.push(marked[pos:pos + n])
            pos += n
            body += c.pull()
        c.flush()
        body += c.pull()
        assert c.sigproc_header() + bytes(body) == got
    # whole-file path: an input with missing frames takes the stream path (gap filling), one with invalid flags only
    # stays on the overlapped path
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for name, data in (("gaps.vdif", marked), ("invalid.vdif", np.concatenate([marked[: 7 * 8032], raw[7 * 8032:]]))):
            path = os.path.join(d, name)
            data.tofile(path)
            oc = pu.oracle_cfg(16.0, 32, 0.012, freq_res=64)
            want = o.channelise(data, oc)
            with ch.Channeliser(cfg, emu_lib) as c:
                c.run_file(path, os.path.join(d, name + ".fil"))
                assert c.get_info().frames_invalid == 1
            pu.check_codes(want, open(os.path.join(d, name + ".fil"), "rb").read(), oc)
    broken = raw.copy()
    broken[7 * 8032 + 8: 7 * 8032 + 11] = 0                        # frame length field zeroed
    with ch.Channeliser(cfg, emu_lib) as c:
        with pytest.raises(ch.RunError):
            c.push(broken)
