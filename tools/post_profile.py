#!/usr/bin/env python3
"""Device-resident runs of the kernels either side of the filterbank, for `rocprofv3 --kernel-trace --stats` (run on the GPU box):
dedispersion of a DM range (64 DMs, 10 s x 1024 channels, 8 bit), the fold of the same rows, the corner turn of a 16-channel
recorder stream.  Prints the bytes each stage moves so that GB/s follow from the profiler's durations."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frb_baseband_amd import _lib, cornerturn as ct, post   # noqa: E402
from tests.hipmem import DeviceBuffer                        # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(3)
nrows, nchan = 312500, 1024
hdr = dict(nchans=nchan, nifs=1, nbits=8, fch1=1416.0 - 0.015625, foff=-0.03125, tsamp=32e-6, tstart=59000.0)
data = rng.integers(100, 156, size=(nrows, nchan), dtype=np.uint8)
d_rows = DeviceBuffer.from_numpy(data)
desc = post.fil_desc(hdr)
err = C.create_string_buffer(256)
out = {"rows_bytes": int(data.nbytes)}

dms = np.asarray(post.dm_list(300.0, 363.0, 1.0), dtype=np.float64)
nout = lib.frbch_dedisperse_nout(C.byref(desc), nrows, dms.ctypes.data, len(dms))
d_out = DeviceBuffer(len(dms) * nout * 4)
nclip = C.c_uint64(0)
for _ in range(5):
    assert lib.frbch_dedisperse_device(C.byref(desc), d_rows.ptr, nrows, dms.ctypes.data, len(dms), 0, 0.0, 0, d_out.ptr, nout,
                                       C.byref(nclip), err, len(err)) == 0, err.value
out["dedisp"] = {"ndm": len(dms), "bytes_in_once": int(data.nbytes), "bytes_in_per_dm_naive": int(data.nbytes) * len(dms),
                 "bytes_out": int(len(dms) * nout * 4)}

nbin, subint = 512, 10.0
nsub = lib.frbch_fold_nsub(C.byref(desc), nrows, C.c_double(subint))
d_prof = DeviceBuffer(nsub * nbin * nchan * 8)
d_hits = DeviceBuffer(nsub * nbin * nchan * 4)
lib.frbch_fold_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                  C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
for _ in range(5):
    assert lib.frbch_fold_device(C.byref(desc), d_rows.ptr, nrows, 1.0 / 0.0334, 0.0, 58999.0, 56.7, 1, nbin, subint, 0, d_prof.ptr, d_hits.ptr,
                                 nsub, err, len(err)) == 0, err.value
out["fold"] = {"bytes_in": int(data.nbytes), "nbin": nbin, "nsub": int(nsub)}

mode = "VDIF_8000-1024-16-2"
fps, recipe, bits = ct.MODES[mode]
hb, pin, _ = ct.frame_geometry(mode)
nfr = 40000                                                   # 321 MB of recorder frames
frames = rng.integers(0, 256, size=nfr * (hb + pin), dtype=np.uint8)
d_in = DeviceBuffer.from_numpy(frames)
info = ct.recipe_info(recipe, lib)
each = nfr * pin * info["bits_per_word"] // info["word_bits"]
outs = [DeviceBuffer(each) for _ in range(info["ntags"])]
ptrs = (C.c_void_p * info["ntags"])(*[b.ptr.value for b in outs])
for _ in range(5):
    assert lib.frbch_cornerturn_device(recipe.encode(), d_in.ptr, nfr, hb + pin, hb, ptrs, info["ntags"], each, 0, err, len(err)) == 0, err.value
out["cornerturn"] = {"mode": mode, "bytes_in": int(frames.nbytes), "bytes_out": int(each * info["ntags"])}
print(json.dumps(out))
