#!/usr/bin/env python3
"""PCIe-inclusive rate of the whole-file path (what the digifil shim does): file -> host -> HBM -> .fil on disk."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frb_baseband_amd import synth, process_vdif as pv
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
d = "/tmp/frbch_rf"; os.makedirs(d, exist_ok=True)
vd = os.path.join(d, "x_ef_no0001_IF1.vdif")
if not os.path.exists(vd):
    one = synth.make_vdif(1.0, bw_mhz=32.0, nchan=1024)
    with open(vd, "wb") as f:
        for i in range(int(secs)):
            f.write(one.tobytes())      # repeated second: fine for a rate measurement
hdr = pv.make_hdr("R3", 1340.49, vd, pol=2, usb=False, ra="01:58:00.75", dec="65:43:00.3", bw=32.0, telescope="effelsberg")
for backend in ("abi", "abi", "shim", "shim"):   # the first call of each kind pays HIP initialisation / page-cache warm-up
    t0 = time.perf_counter()
    out = pv.run_digifil(hdr, d, 0, secs, 1024, overwrite=True, pol=2, nbit=8, backend=backend)
    dt = time.perf_counter() - t0
    n = os.path.getsize(out)
    print(f"{backend}: {secs} s of 32 MHz IF in {dt:.3f} s -> {secs*64e6/dt/1e6:.1f} Msamples/s ({secs/dt:.1f}x real time), {n} bytes out")
