"""MI355X-native VDIF -> SIGPROC-filterbank channeliser (drop-in for the ``digifil`` call of
pharaofranz/frb-baseband's process_vdif.py:157-191).

Layout:
  csrc/            hand-written HIP kernels (gfx950) + the C-ABI shared library (include/frbch.h)
  _lib.py          ctypes binding of that library (fails loudly when it is missing)
  channeliser.py   Python object over the C-ABI handle
  digifil_args.py  parser for the digifil argv the reference builds
  process_vdif.py  host-side mirror of the reference's per-IF harness (same CLI, same names)
  multi_if.py      IF sharding over GPUs/ranks + host-side frequency concatenation (splice)
  vdif.py, sigproc.py, synth.py   format helpers and the seeded synthetic generator
"""
__version__ = "0.1.0"
