"""Tiny ctypes view of the HIP runtime for tests: device buffers without torch."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


class DeviceBuffer:
    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        rc = hip().hipMalloc(C.byref(self.ptr), max(1, nbytes))
        if rc != 0:
            raise RuntimeError(f"hipMalloc failed ({rc})")

    @classmethod
    def from_numpy(cls, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        rc = hip().hipMemcpy(buf.ptr, arr.ctypes.data, arr.nbytes, 1)  # hipMemcpyHostToDevice
        assert rc == 0
        return buf

    def to_numpy(self, dtype, count=None) -> np.ndarray:
        hip().hipDeviceSynchronize()
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize if count is None else count, dtype=dtype)
        rc = hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2)  # hipMemcpyDeviceToHost
        assert rc == 0
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
