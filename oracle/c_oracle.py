"""ctypes view of the plain-C oracle port (oracle/frb_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libfrb_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = C.CDLL(_PATH)
        lib.frbo_plan_create.restype = C.c_void_p
        lib.frbo_plan_create.argtypes = [C.c_size_t, C.c_size_t]
        lib.frbo_plan_destroy.argtypes = [C.c_void_p]
        lib.frbo_block_power.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = lib
    return _lib


def available() -> bool:
    try:
        _load()
        return True
    except Exception:
        return False


def block_power(payload: np.ndarray, nchan: int, freq_res: int, nblocks: int, pol_mode: int = 2, tscr: int = 1):
    """payload bytes (headers stripped, whole blocks) -> float32 [nif][C][nblocks * R/tscr]."""
    lib = _load()
    n = 2 * nchan * freq_res
    nif = 4 if pol_mode >= 4 else 1
    nt = freq_res // tscr
    plan = lib.frbo_plan_create(nchan, freq_res)
    out = np.empty((nif, nchan, nblocks * nt), dtype=np.float32)
    tmp = np.empty((nif, nchan, nt), dtype=np.float32)
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    try:
        for b in range(nblocks):
            blk = payload[b * n // 2:(b + 1) * n // 2]
            lib.frbo_block_power(plan, blk.ctypes.data, pol_mode, tscr, tmp.ctypes.data)
            out[:, :, b * nt:(b + 1) * nt] = tmp
    finally:
        lib.frbo_plan_destroy(plan)
    return out


def channelise_blocks(raw_frames: np.ndarray, bw: float, nchan: int, freq_res: int, nblocks: int, pol_mode: int = 2):
    """Frames -> power of the first nblocks; the timed body of bench.py's cpu_baseline."""
    fb = 8032
    payload = raw_frames[: raw_frames.size // fb * fb].reshape(-1, fb)[:, 32:].reshape(-1)
    return block_power(payload, nchan, freq_res, nblocks, pol_mode)


def _bench_worker(seconds: float, bw: float, nchan: int, seed: int = 0, pol_mode: int = 2) -> float:
    """one `digifil_nthreads=1` process of the reference's per-IF fan-out (base2fil.sh:60-66,219): blocks/s"""
    import time
    from frb_baseband_amd import synth
    from oracle import frb_oracle as o
    r = o.freq_res_for(nchan)
    n = 2 * nchan * r
    raw = synth.make_vdif(2 * n / (2e6 * bw) + 0.001, bw_mhz=bw, nchan=nchan, if_index=seed)
    t0 = time.perf_counter()
    nblk = 0
    while time.perf_counter() - t0 < seconds:
        channelise_blocks(raw, bw, nchan, r, 2, pol_mode)
        nblk += 2
    return nblk * n / (time.perf_counter() - t0)


if __name__ == "__main__":   # python -m oracle.c_oracle <seconds> <bw> <nchan> <seed> [pol_mode]: prints samples/s (bench.py's all-cores leg)
    import sys
    print(_bench_worker(float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]),
                        int(sys.argv[5]) if len(sys.argv) > 5 else 2))
