"""Python object over one C-ABI handle (= one IF, like one digifil process, base2fil.sh:60-66)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class Error(Exception):
    """Base class of this module's exceptions (process_vdif.py:227-229): ``except process_vdif.Error`` catches both."""


class InputError(Error):
    """Bad input / unsupported configuration (same class name and base as process_vdif.py:232-241)."""

    def __init__(self, message):
        super().__init__(message)
        self.message = message


class RunError(Error):
    """The channeliser failed while running (process_vdif.py:244-253)."""

    def __init__(self, message):
        super().__init__(message)
        self.message = message


def new_config(lib=None, **kw) -> _lib.FrbchConfig:
    lib = lib or _lib.load()
    cfg = _lib.FrbchConfig()
    lib.frbch_config_init(C.byref(cfg))
    for key, val in kw.items():
        if not hasattr(cfg, key):
            raise InputError(f"unknown configuration field {key!r}")
        if isinstance(val, str):
            val = val.encode()
        if key == "levels":
            val = (C.c_float * 4)(*[float(x) for x in val])
        setattr(cfg, key, val)
    return cfg


class Channeliser:
    """``with Channeliser(cfg) as ch: ...``"""

    def __init__(self, cfg: _lib.FrbchConfig, lib=None):
        self.lib = lib or _lib.load()
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self.lib.frbch_open(C.byref(cfg), C.byref(self._h))
        if rc != _lib.OK:
            msg = self._err(rc)
            self.close()
            raise (InputError if rc == _lib.E_ARG else RunError)(msg)
        self.info = self.get_info()

    # -- plumbing -------------------------------------------------------------------------------
    def _err(self, rc: int) -> str:
        detail = self.lib.frbch_last_error(self._h).decode() if self._h else ""
        return f"{self.lib.frbch_strerror(rc).decode()}: {detail}"

    def _check(self, rc: int):
        if rc < 0:
            raise (InputError if rc == _lib.E_ARG else RunError)(self._err(rc))
        return rc

    def close(self):
        if self._h:
            self.lib.frbch_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_info(self) -> _lib.FrbchInfo:
        info = _lib.FrbchInfo()
        self._check(self.lib.frbch_get_info(self._h, C.byref(info)))
        return info

    def reset(self) -> None:
        self._check(self.lib.frbch_reset(self._h))

    # -- whole file (what `digifil ... -o out hdr` does) ----------------------------------------
    def run_file(self, vdif_path: str, out_fil: str):
        self._check(self.lib.frbch_run_file(self._h, vdif_path.encode(), out_fil.encode()))
        self.info = self.get_info()

    # -- host streaming ---------------------------------------------------------------------------
    def push(self, frames) -> None:
        buf = np.ascontiguousarray(np.frombuffer(frames, dtype=np.uint8) if not isinstance(frames, np.ndarray) else frames)
        self._check(self.lib.frbch_push(self._h, buf.ctypes.data, buf.size))

    def flush(self) -> None:
        self._check(self.lib.frbch_flush(self._h))

    def pull(self, max_bytes: int = 1 << 26) -> bytes:
        out = bytearray()
        tmp = np.empty(min(max_bytes, 1 << 24), dtype=np.uint8)
        while len(out) < max_bytes:
            n = self._check(self.lib.frbch_pull(self._h, tmp.ctypes.data, min(tmp.size, max_bytes - len(out))))
            if n == 0:
                break
            out += tmp[:n].tobytes()
        return bytes(out)

    def sigproc_header(self) -> bytes:
        tmp = np.empty(4096, dtype=np.uint8)
        n = self._check(self.lib.frbch_sigproc_header(self._h, tmp.ctypes.data, tmp.size))
        return tmp[:n].tobytes()

    def channelise_bytes(self, frames) -> bytes:
        """frames -> complete .fil bytes (header + samples) through the streaming path."""
        self.push(frames)
        self.flush()
        body = bytearray()
        while True:
            chunk = self.pull()
            if not chunk:
                break
            body += chunk
        return self.sigproc_header() + bytes(body)

    # -- device-resident path ---------------------------------------------------------------------
    def process_device(self, d_frames_ptr: int, nframes: int, frame_bytes: int, header_bytes: int,
                       payload_off: int, nblocks: int, d_out_ptr: int, out_cap: int, stream: int = 0) -> int:
        rows = C.c_uint64(0)
        self._check(self.lib.frbch_process_device(self._h, d_frames_ptr, nframes, frame_bytes, header_bytes,
                                                  payload_off, nblocks, d_out_ptr, out_cap, C.byref(rows),
                                                  stream or None))
        return rows.value

    def flush_device(self, d_out_ptr: int, out_cap: int, stream: int = 0) -> int:
        rows = C.c_uint64(0)
        self._check(self.lib.frbch_flush_device(self._h, d_out_ptr, out_cap, C.byref(rows), stream or None))
        return rows.value

    def power_device(self, d_frames_ptr: int, nframes: int, frame_bytes: int, header_bytes: int,
                     payload_off: int, nblocks: int, d_power_ptr: int, cap: int, stream: int = 0) -> None:
        self._check(self.lib.frbch_power_device(self._h, d_frames_ptr, nframes, frame_bytes, header_bytes,
                                                payload_off, nblocks, d_power_ptr, cap, stream or None))

    def unpack_device(self, d_frames_ptr: int, nframes: int, frame_bytes: int, header_bytes: int, payload_off: int,
                      nsamples: int, decoder: int, d_volt_ptr: int, cap: int, stream: int = 0) -> None:
        self._check(self.lib.frbch_unpack_device(self._h, d_frames_ptr, nframes, frame_bytes, header_bytes, payload_off,
                                                 nsamples, decoder, d_volt_ptr, cap, stream or None))

    # -- rescale state ---------------------------------------------------------------------------
    def get_rescale(self):
        n = self.info.nif * self.info.nchan
        off = np.empty(n, dtype=np.float32)
        sc = np.empty(n, dtype=np.float32)
        self._check(self.lib.frbch_get_rescale(self._h, off.ctypes.data, sc.ctypes.data))
        shape = (self.info.nif, self.info.nchan)
        return off.reshape(shape), sc.reshape(shape)

    def set_rescale(self, offset, scale) -> None:
        off = np.ascontiguousarray(offset, dtype=np.float32).reshape(-1)
        sc = np.ascontiguousarray(scale, dtype=np.float32).reshape(-1)
        assert off.size == sc.size == self.info.nif * self.info.nchan
        self._check(self.lib.frbch_set_rescale(self._h, off.ctypes.data, sc.ctypes.data))

    # -- measurement -------------------------------------------------------------------------------
    def set_profiling(self, on: bool) -> None:
        self._check(self.lib.frbch_set_profiling(self._h, 1 if on else 0))

    def timing_reset(self) -> None:
        self._check(self.lib.frbch_timing_reset(self._h))

    def get_timing(self) -> dict:
        t = _lib.FrbchTiming()
        self._check(self.lib.frbch_get_timing(self._h, C.byref(t)))
        return {t.k[i].name.decode(): {"launches": t.k[i].launches, "total_ms": t.k[i].total_ms,
                                       "algorithmic_bytes": t.k[i].algorithmic_bytes}
                for i in range(t.nkernels)}
