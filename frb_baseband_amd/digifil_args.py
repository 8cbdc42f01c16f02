"""digifil command line -> channeliser configuration.

The parsing itself is the C-ABI function ``frbch_parse_digifil_argv`` (one implementation shared
with the ``digifil`` CLI shim); this module only splits the string run_digifil builds
(process_vdif.py:156-182) and reads the .hdr it names (process_vdif.py:115-139).
"""
from __future__ import annotations

import ctypes as C
import shlex

from . import _lib
from .channeliser import InputError, new_config


def parse(cmd, lib=None, read_hdr: bool = True):
    """``cmd``: the digifil command string or argv list (argv[0] = program name).
    Returns (cfg, hdr_path, out_path)."""
    lib = lib or _lib.load()
    argv = shlex.split(cmd) if isinstance(cmd, str) else list(cmd)
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    cfg = new_config(lib)
    hdr = C.create_string_buffer(1024)
    out = C.create_string_buffer(1024)
    err = C.create_string_buffer(512)
    rc = lib.frbch_parse_digifil_argv(len(argv), arr, C.byref(cfg), hdr, len(hdr), out, len(out), err, len(err))
    if rc != _lib.OK:
        raise InputError(f"{lib.frbch_strerror(rc).decode()}: {err.value.decode()}")
    hdr_path, out_path = hdr.value.decode(), out.value.decode()
    if read_hdr:
        rc = lib.frbch_config_from_hdr(hdr_path.encode(), C.byref(cfg))
        if rc != _lib.OK:
            raise InputError(f"cannot use header {hdr_path}: {lib.frbch_strerror(rc).decode()}")
    return cfg, hdr_path, out_path
