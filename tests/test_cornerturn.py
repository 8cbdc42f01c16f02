"""Corner turn: the recorder stream split into per-IF streams on the GPU, against the numpy restatement
(bit-exact), for every mode of spif2file.sh:31-113, through the TEST-ONLY emulator build; the GPU runs are in
test_gpu_post-style below (marked gpu)."""
import numpy as np
import pytest

from frb_baseband_amd import channeliser as ch, cornerturn as ct, sigproc, synth, vdif
from oracle import frb_oracle as o, post_oracle as po
from tests import parity_util as pu


def recorder_frames(mode, nframes, seed=0):
    hb, pin, _ = ct.frame_geometry(mode)
    rng = np.random.default_rng(seed)
    fr = rng.integers(0, 256, size=(nframes, hb + pin), dtype=np.uint8)
    return fr.reshape(-1), hb, pin


@pytest.mark.parametrize("mode", sorted(ct.MODES))
def test_every_mode_matches_the_oracle_on_the_emulator(emu_lib, mode):
    fps, recipe, bits = ct.MODES[mode]
    frames, hb, pin = recorder_frames(mode, 4)
    got = ct.split_host(frames, recipe, hb + pin, hb, lib=emu_lib)
    payload = frames.reshape(4, hb + pin)[:, hb:].reshape(-1)
    want = po.cornerturn(payload, recipe)
    info = ct.recipe_info(recipe, emu_lib)
    assert len(got) == len(want) == info["ntags"]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # 2 pols x bits per sample per IF and word: the per-IF stream carries 2 channels
    assert info["bits_per_word"] == 2 * bits
    assert np.array_equal(po.interleave(want, recipe)[: payload.size] | 0, payload) or "16-2-2" in mode or info["ntags"] * info["bits_per_word"] < info["word_bits"]


def test_recipe_errors_and_flip(emu_lib):
    for bad in ("32[0,1]", "33>[0,1,2,3]:0", "32>[0,1,2,40]:0", "32>[0,1,2,3][4,5]:0-1", "8>[0,1,2,3][4,5,6,7]:0-2"):
        with pytest.raises(ct.InputError):
            ct.recipe_info(bad, emu_lib)
    r = ct.MODES["VDIF_8000-1024-8-2"][1]
    assert ct.flip_recipe(r, 4) == "16>[0,1,4,5][8,9,12,13][2,3,6,7][10,11,14,15]:0-3"
    assert ct.frame_geometry("MARK5B-2048-16-2") == (16, 10000, 10000) and ct.frame_geometry("VDIF_1000-1024-16-2") == (32, 1000, 1000)


def _eight_if_recorder(secs, bw, nchan):
    """8 per-IF 2-bit streams (the synthetic IFs of the other tests) interleaved into one 16-channel recorder stream"""
    mode = "VDIF_8000-1024-16-2"
    recipe = ct.MODES[mode][1]
    per_if = []
    for i in range(8):
        raw = synth.make_vdif(secs, bw_mhz=bw, nchan=nchan, if_index=i + 1)
        per_if.append(o.strip_frames(raw, 8032, 32))
    rec = po.interleave(per_if, recipe)
    nfr = rec.size // 8000
    frames = vdif.frame_payload(rec[: nfr * 8000], bw_mhz=bw * 8, payload_bytes=8000)   # (header fields are not read by the split)
    return frames, recipe, per_if


def test_split_streams_feed_the_channeliser_in_place(emu_lib):
    """recorder stream -> GPU split -> frbch_process_device on the tag's payload (header_bytes = 0): the same .fil bytes
    as the per-IF VDIF file through the normal path"""
    frames, recipe, per_if = _eight_if_recorder(0.02, 16.0, 32)
    outs = ct.split_host(frames, recipe, 8032, 32, lib=emu_lib)
    for i in (0, 5):
        n = outs[i].size
        assert np.array_equal(outs[i], per_if[i][:n])
    cfg = pu.lib_cfg(emu_lib, 16.0, 32, 0.02, freq_res=64)
    with ch.Channeliser(cfg, emu_lib) as c:
        info = c.info
        stream = np.ascontiguousarray(outs[5])
        nfr = stream.size // 8000
        nblocks = (nfr * 8000) // info.block_payload_bytes
        rows = nblocks * info.rows_per_block
        out = np.zeros(rows * info.row_bytes, np.uint8)
        r1 = c.process_device(stream.ctypes.data, nfr, 8000, 0, 0, nblocks, out.ctypes.data, out.size)
        r2 = c.flush_device(out.ctypes.data + r1 * info.row_bytes, out.size - r1 * info.row_bytes)
        assert r1 + r2 == rows
    with ch.Channeliser(cfg, emu_lib) as c:
        ref = sigproc.read_fil(c.channelise_bytes(synth.make_vdif(0.02, bw_mhz=16.0, nchan=32, if_index=6)))
    assert np.array_equal(out.reshape(ref.data.shape)[: rows], ref.data[: rows])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(ct.MODES))
def test_modes_match_the_oracle_on_the_gpu(hip_lib, mode):
    fps, recipe, bits = ct.MODES[mode]
    frames, hb, pin = recorder_frames(mode, 64, seed=5)
    got = ct.split_host(frames, recipe, hb + pin, hb, lib=hip_lib)
    want = po.cornerturn(frames.reshape(64, hb + pin)[:, hb:].reshape(-1), recipe)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.gpu
def test_recorder_stream_to_filterbank_without_per_if_files(hip_lib):
    """16-channel recorder stream in HBM -> frbch_cornerturn_device -> frbch_process_device per IF, nothing leaves the
    card in between; config-2 shape for two of the eight IFs, against the oracle"""
    import ctypes as C
    from tests.hipmem import DeviceBuffer
    frames, recipe, per_if = _eight_if_recorder(0.14, 32.0, 1024)
    d_in = DeviceBuffer.from_numpy(frames)
    nfr = frames.size // 8032
    each = nfr * 8000 * 4 // 32
    outs = [DeviceBuffer(each) for _ in range(8)]
    ptrs = (C.c_void_p * 8)(*[b.ptr.value for b in outs])
    err = C.create_string_buffer(256)
    rc = hip_lib.frbch_cornerturn_device(recipe.encode(), d_in.ptr, nfr, 8032, 32, ptrs, 8, each, 0, err, len(err))
    assert rc == 0, err.value
    for i in (0, 7):
        assert np.array_equal(outs[i].to_numpy(np.uint8), per_if[i][:each])
        raw = synth.make_vdif(0.14, bw_mhz=32.0, nchan=1024, if_index=i + 1)
        ocfg = pu.oracle_cfg(32.0, 1024, 0.14)
        ref = o.channelise(raw, ocfg)
        with ch.Channeliser(pu.lib_cfg(hip_lib, 32.0, 1024, 0.14), hip_lib) as c:
            info = c.info
            nf = each // 8000
            nblocks = (nf * 8000) // info.block_payload_bytes
            rows = nblocks * info.rows_per_block
            out = DeviceBuffer(rows * info.row_bytes)
            r1 = c.process_device(outs[i].ptr.value, nf, 8000, 0, 0, nblocks, out.ptr.value, out.nbytes)
            r2 = c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
            assert r1 + r2 == rows
            body = out.to_numpy(np.uint8).tobytes()
        want = sigproc.read_fil(ref)
        got = np.frombuffer(body, np.uint8).reshape(rows, 1, 1024)
        d = got.astype(int) - want.data[:rows].astype(int)
        assert np.abs(d).max() <= 1 and np.count_nonzero(d) <= 2e-4 * d.size
