#!/usr/bin/env python3
"""Host-inclusive scan path in context (run on the GPU box): frbch_run_scan of N IFs (VDIF files on tmpfs) into /dev/null, a
drained FIFO and a tmpfs file, with the phase clock of the profiling build (FRBCH_LIB=.../libfrbch_exp.so FRBCH_TIMING=1),
next to the PCIe ceilings of the box measured with pinned buffers (torch)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frb_baseband_amd import channeliser as ch, multi_if, synth
nif = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pol = int(sys.argv[2]) if len(sys.argv) > 2 else 5
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
try:
    import torch
    n = 1 << 30
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); t = time.perf_counter() - t0
        print("PCIe %s, 1 GiB pinned: %.1f ms = %.1f GB/s" % (name, t * 1e3, n / t / 1e9))
    del h, d
except Exception as exc:
    print("torch PCIe probe skipped:", exc)
vds = []
for i in range(nif):
    raw = synth.make_vdif(secs, bw_mhz=32.0, nchan=1024, if_index=i + 1)
    p = f"/dev/shm/frbch_scan_t_if{i}.vdif"
    raw.tofile(p)
    vds.append(p)
nbytes_in = raw.size * nif
chans = [ch.Channeliser(ch.new_config(bw_mhz=(-32.0 if (nif - i) % 2 else 32.0), nchan=1024, pol_mode=pol, total_s=secs, rescale_constant=1)) for i in range(nif)]
info = chans[0].info
out_bytes = nif * info.row_bytes * (int(secs * 64e6) // (2 * 1024) )
def run(sink):
    for c in chans:
        c.reset()
    multi_if.run_scan(chans, vds, sink)
def timed(sink, n=3, before=None, after=None):
    ts = []
    for _ in range(n):
        if before: before()
        t0 = time.perf_counter(); run(sink); 
        if after: after()
        ts.append(time.perf_counter() - t0)
    return min(ts)
run("/dev/null")
t = timed("/dev/null"); print("frbch_run_scan %d IFs pol %d -> /dev/null: %.1f ms = %.2f Gsamples/s (in %.2f GB, out ~%.2f GB)" % (nif, pol, t * 1e3, nbytes_in * 2 / t / 1e9, nbytes_in / 1e9, out_bytes / 1e9))
fifo = "/dev/shm/frbch_scan_t.fifo"
if os.path.exists(fifo): os.remove(fifo)
os.mkfifo(fifo)
th = [None]
def start_drain():
    def drain():
        with open(fifo, "rb", buffering=0) as f:
            buf = bytearray(1 << 24)
            while f.readinto(buf): pass
    th[0] = threading.Thread(target=drain); th[0].start()
t = timed(fifo, 2, before=start_drain, after=lambda: th[0].join()); print("frbch_run_scan -> FIFO (drained): %.1f ms = %.2f Gsamples/s" % (t * 1e3, nbytes_in * 2 / t / 1e9))
for c in chans:
    c.close()
for f in vds + [fifo]:
    if os.path.exists(f): os.remove(f)
