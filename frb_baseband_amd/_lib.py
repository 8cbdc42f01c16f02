"""ctypes binding of the C-ABI library (include/frbch.h).

The product path is the HIP library ``csrc/libfrbch.so``.  If it is missing or cannot be loaded
this module raises -- there is no Python/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRBCH_LIB: another build of the same library for this process (the profiling build `make exp`); default: the product
LIB_PATH = os.environ.get("FRBCH_LIB") or os.path.join(_HERE, "csrc", "libfrbch.so")

ABI_VERSION = 5

OK, E_ARG, E_IO, E_FORMAT, E_DEVICE, E_NOMEM, E_STATE, E_CAPACITY = 0, -1, -2, -3, -4, -5, -6, -7


class FrbchConfig(C.Structure):
    _fields_ = [
        ("size", C.c_uint32), ("abi_version", C.c_uint32),
        ("freq_mhz", C.c_double), ("bw_mhz", C.c_double),
        ("start_s", C.c_double), ("total_s", C.c_double),
        ("nchan", C.c_uint32), ("freq_res", C.c_uint32), ("tscrunch", C.c_uint32),
        ("nbit_out", C.c_int32), ("pol_mode", C.c_int32), ("rescale_constant", C.c_uint32),
        ("rescale_interval_s", C.c_double), ("dm", C.c_double),
        ("coherent", C.c_uint32), ("device", C.c_int32),
        ("max_blocks_per_launch", C.c_uint32), ("flags", C.c_uint32),
        ("telescope", C.c_char * 64), ("source", C.c_char * 64),
        ("ra", C.c_char * 32), ("dec", C.c_char * 32), ("datafile", C.c_char * 512),
        ("input_bits", C.c_uint32), ("overlap", C.c_uint32),
        ("levels", C.c_float * 4),
        ("unpack_mode", C.c_uint32), ("dls_nsample", C.c_uint32),
        ("dls_cutoff_sigma", C.c_float), ("dls_threshold", C.c_float),
    ]


class FrbchInfo(C.Structure):
    _fields_ = [
        ("size", C.c_uint32), ("nchan", C.c_uint32), ("freq_res", C.c_uint32),
        ("tscrunch", C.c_uint32), ("nif", C.c_uint32),
        ("block_samples", C.c_uint64), ("block_payload_bytes", C.c_uint64),
        ("rows_per_block", C.c_uint64), ("row_bytes", C.c_uint64),
        ("rescale_interval_rows", C.c_uint64), ("rows_out", C.c_uint64),
        ("blocks_done", C.c_uint64),
        ("tsamp_s", C.c_double), ("tstart_mjd", C.c_double),
        ("fch1_mhz", C.c_double), ("foff_mhz", C.c_double),
        ("frame_bytes", C.c_uint32), ("header_bytes", C.c_uint32),
        ("have_rescale", C.c_uint32), ("diag", C.c_uint32),
        ("frames_seen", C.c_uint64), ("frames_invalid", C.c_uint64), ("frame_gaps", C.c_uint64),
        ("block_stride_bytes", C.c_uint64), ("nfilt_pos", C.c_uint32), ("nfilt_neg", C.c_uint32),
        ("frames_filled", C.c_uint64),
    ]


class FrbchFilDesc(C.Structure):
    _fields_ = [("size", C.c_uint32), ("nchan", C.c_uint32), ("nifs", C.c_uint32), ("nbits", C.c_int32),
                ("product", C.c_uint32), ("reserved", C.c_uint32),
                ("fch1_mhz", C.c_double), ("foff_mhz", C.c_double), ("tsamp_s", C.c_double), ("tstart_mjd", C.c_double)]


class _KTiming(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("algorithmic_bytes", C.c_double)]


class FrbchTiming(C.Structure):
    _fields_ = [("size", C.c_uint32), ("nkernels", C.c_uint32), ("k", _KTiming * 12)]


# every symbol include/frbch.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "frbch_config_init": (C.c_int, [C.POINTER(FrbchConfig)]),
    "frbch_config_from_hdr": (C.c_int, [C.c_char_p, C.POINTER(FrbchConfig)]),
    "frbch_parse_digifil_argv": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(FrbchConfig),
                                           C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                           C.c_char_p, C.c_size_t]),
    "frbch_open": (C.c_int, [C.POINTER(FrbchConfig), C.POINTER(_P)]),
    "frbch_close": (None, [_P]),
    "frbch_last_error": (C.c_char_p, [_P]),
    "frbch_strerror": (C.c_char_p, [C.c_int]),
    "frbch_get_info": (C.c_int, [_P, C.POINTER(FrbchInfo)]),
    "frbch_reset": (C.c_int, [_P]),
    "frbch_run_file": (C.c_int, [_P, C.c_char_p, C.c_char_p]),
    "frbch_run_scan": (C.c_int, [C.POINTER(_P), C.c_uint32, C.POINTER(C.c_char_p), C.c_char_p]),
    "frbch_scan_device": (C.c_int, [C.POINTER(_P), C.c_uint32, C.POINTER(_P), C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
                                    C.c_uint64, C.c_int, _P, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64), _P]),
    "frbch_push": (C.c_int, [_P, _P, C.c_size_t]),
    "frbch_flush": (C.c_int, [_P]),
    "frbch_pull": (C.c_long, [_P, _P, C.c_size_t]),
    "frbch_sigproc_header": (C.c_long, [_P, _P, C.c_size_t]),
    "frbch_process_device": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.c_uint64, _P, C.c_size_t, C.POINTER(C.c_uint64), _P]),
    "frbch_flush_device": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_uint64), _P]),
    "frbch_power_device": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
                                     C.c_uint64, _P, C.c_size_t, _P]),
    "frbch_unpack_device": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                      _P, C.c_size_t, _P]),
    "frbch_get_rescale": (C.c_int, [_P, _P, _P]),
    "frbch_set_rescale": (C.c_int, [_P, _P, _P]),
    "frbch_dedisperse_nout": (C.c_long, [C.POINTER(FrbchFilDesc), C.c_uint64, _P, C.c_uint32]),
    "frbch_dedisperse_host": (C.c_int, [C.POINTER(FrbchFilDesc), _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, C.c_double,
                                        C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "frbch_dedisperse_device": (C.c_int, [C.POINTER(FrbchFilDesc), _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, C.c_double,
                                          C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "frbch_fold_nsub": (C.c_long, [C.POINTER(FrbchFilDesc), C.c_uint64, C.c_double]),
    "frbch_fold_host": (C.c_int, [C.POINTER(FrbchFilDesc), _P, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_uint32, C.c_uint32, C.c_double, C.c_int, _P, _P, C.c_uint32, C.c_char_p, C.c_size_t]),
    "frbch_fold_device": (C.c_int, [C.POINTER(FrbchFilDesc), _P, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_uint32, C.c_uint32, C.c_double, C.c_int, _P, _P, C.c_uint32, C.c_char_p, C.c_size_t]),
    "frbch_cornerturn_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]),
    "frbch_cornerturn_host": (C.c_int, [C.c_char_p, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(_P), C.c_uint32,
                                        C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]),
    "frbch_cornerturn_device": (C.c_int, [C.c_char_p, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(_P), C.c_uint32,
                                          C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]),
    "frbch_set_profiling": (C.c_int, [_P, C.c_int]),
    "frbch_timing_reset": (C.c_int, [_P]),
    "frbch_get_timing": (C.c_int, [_P, C.POINTER(FrbchTiming)]),
    "frbch_version": (C.c_char_p, []),
}

_cache: dict = {}


class LibraryMissing(RuntimeError):
    pass


def load(path: str | None = None) -> C.CDLL:
    """Load the C-ABI library and bind every declared symbol.  Raises LibraryMissing loudly."""
    path = path or LIB_PATH
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as "
            f"g; g.build()' or make -C frb_baseband_amd/csrc).  There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # missing libamdhip64 etc.
        raise LibraryMissing(f"cannot load {path}: {exc}.  There is no CPU fallback.") from exc
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
