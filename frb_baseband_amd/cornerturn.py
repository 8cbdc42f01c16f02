"""The corner turn in front of the channeliser, on the GPU.

The reference asks jive5ab (``spif2file``) to split the recorder's VDIF / Mark5B stream -- all IFs interleaved in every
W-bit word -- into one 2-channel x 2-bit VDIF file per IF (spif2file.sh:178-186), with a per-mode recipe string
(spif2file.sh:31-113).  ``MODES`` restates that table (mode -> frames per second, recipe, bits per sample);
``flip_recipe`` is the `flipped` re-ordering of spif2file.sh:116-131.  ``split_device`` performs the split in HBM through
the C ABI, so that ``frbch_process_device`` can read each IF's payload stream in place (``header_bytes = 0``) and the
per-IF files are never written; ``split_to_files`` writes them for pipelines that still want them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, vdif
from .channeliser import InputError, RunError

_R32A = "32>[16,17,24,25][0,1,8,9][18,19,26,27][2,3,10,11][20,21,28,29][4,5,12,13][22,23,30,31][6,7,14,15]:0-7"
_R32B = "32>[24,25,16,17][8,9,0,1][26,27,18,19][10,11,2,3][28,29,20,21][12,13,4,5][30,31,22,23][14,15,6,7]:0-7"
_R64 = ("64>[16,17,48,49][0,1,32,33][18,19,50,51][2,3,34,35][20,21,52,53][4,5,36,37][22,23,54,55][6,7,38,39]"
        "[24,25,56,57][8,9,40,41][26,27,58,59][10,11,42,43][28,29,60,61][12,13,44,45][30,31,62,63][14,15,46,47]:0-15")
_R16 = "16>[8,9,12,13][0,1,4,5][10,11,14,15][2,3,6,7]:0-3"

# mode -> (frames per second, recipe, bits per sample)      spif2file.sh:31-113
MODES = {
    "VDIF_8000-4096-32-2": (64000, _R64, 2),
    "VDIF_8000-2048-32-2": (32000, _R64, 2),
    "VDIF_8000-2048-16-2": (32000, _R32A, 2),
    "VDIF_8000-1024-16-2": (16000, _R32B, 2),
    "VDIF_1000-1024-16-2": (128000, _R32B, 2),
    "VDIF_8000-1024-8-2": (16000, _R16, 2),
    "VDIF_8000-1024-16-1": (16000, "16>[8,12][0,4][9,13][1,5][10,14][2,6][11,15][3,7]:0-7", 1),
    "VDIF_8000-512-4-2": (8000, "8>[4,5,6,7][0,1,2,3]:0-1", 2),
    "VDIF_8000-16-2-2": (250, "4>[0,1,2,3]:0", 2),
    "VDIF_8000-32-4-2": (500, "8>[0,1,4,5][2,3,6,7]:0-1", 2),
    "VDIF_8000-512-16-2": (8000, _R32A, 2),
    "MARK5B-1024-16-2": (12800, "swap_sign_mag+" + _R32A, 2),
    "MARK5B-1024-8-2": (12800, "swap_sign_mag+" + _R16, 2),
    "MARK5B-2048-16-2": (25600, "swap_sign_mag+" + _R32A, 2),
    "MARK5B-2048-32-2": (25600, "swap_sign_mag+" + _R64, 2),
}


def frame_geometry(mode: str):
    """(input header bytes, input payload bytes, output payload bytes)   spif2file.sh:100-113"""
    if mode.startswith("VDIF"):
        payload = int(mode[5:9])
        return 32, payload, payload
    if mode.startswith("MARK5B"):
        return 16, 10000, 10000
    raise InputError(f"Cannot determine frame sizes from {mode}.")


def flip_recipe(recipe: str, nif: int) -> str:
    """the `flipped` re-ordering (spif2file.sh:116-131): odd IFs take the bracket of the next IF, even IFs of the previous"""
    head, rest = recipe.split(">", 1)
    groups = rest.split("]")
    br = [g + "]" for g in groups[:-1]]
    tail = groups[-1]
    out, add = [], 1
    for i in range(1, nif + 1):
        out.append(br[i - 1 + add])
        add = -add
    return head + ">" + "".join(out) + tail


def recipe_info(recipe: str, lib=None):
    lib = lib or _lib.load()
    w, n, b, t0 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    err = C.create_string_buffer(256)
    rc = lib.frbch_cornerturn_info(recipe.encode(), C.byref(w), C.byref(n), C.byref(b), C.byref(t0), err, len(err))
    if rc:
        raise InputError(f"{lib.frbch_strerror(rc).decode()}: {err.value.decode()}")
    return dict(word_bits=w.value, ntags=n.value, bits_per_word=b.value, first_tag=t0.value)


def split_host(frames: np.ndarray, recipe: str, frame_bytes: int, header_bytes: int, device: int = 0, lib=None):
    """frames (uint8, whole recorder frames) -> list of per-tag payload byte arrays, through the GPU"""
    lib = lib or _lib.load()
    info = recipe_info(recipe, lib)
    nfr = frames.size // frame_bytes
    nwords = nfr * (frame_bytes - header_bytes) * 8 // info["word_bits"]
    each = nwords * info["bits_per_word"] // 8
    outs = [np.empty(each, np.uint8) for _ in range(info["ntags"])]
    ptrs = (C.c_void_p * info["ntags"])(*[o.ctypes.data for o in outs])
    err = C.create_string_buffer(256)
    buf = np.ascontiguousarray(frames[: nfr * frame_bytes])
    rc = lib.frbch_cornerturn_host(recipe.encode(), buf.ctypes.data, nfr, frame_bytes, header_bytes, ptrs, info["ntags"], each,
                                   device, err, len(err))
    if rc:
        raise (InputError if rc == _lib.E_ARG else RunError)(f"{lib.frbch_strerror(rc).decode()}: {err.value.decode()}")
    return outs


def split_to_files(frames: np.ndarray, mode: str, bw_mhz: float, names, *, seconds0: int = 0, ref_epoch: int = 0, frame0: int = 0,
                   flipped: bool = False, device: int = 0, lib=None):
    """what `spif2file=connect:...:<recipe>=<dir>/if_{tag}` leaves behind (spif2file.sh:178-186): one 2-channel VDIF file
    per IF with `vdifsize` payload bytes per frame.  ``names[i]`` is the path of tag i."""
    fps, recipe, bits = MODES[mode]
    hb, pin, pout = frame_geometry(mode)
    if flipped:
        recipe = flip_recipe(recipe, recipe_info(recipe, lib)["ntags"])
    outs = split_host(frames, recipe, hb + pin, hb, device=device, lib=lib)
    for path, payload in zip(names, outs):
        n = payload.size // pout * pout
        vdif.frame_payload(payload[:n], bw_mhz=bw_mhz, seconds0=seconds0, ref_epoch=ref_epoch, frame0=frame0,
                           payload_bytes=pout, bits=bits).tofile(path)
    return outs
