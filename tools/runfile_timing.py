#!/usr/bin/env python3
"""Host-inclusive path in context: frbch_run_file for 10 s of a 32 MHz IF (VDIF on tmpfs) into (a) a tmpfs file, (b) /dev/null,
(c) a FIFO drained by a reader thread, next to the OS ceilings of the same box: one write() stream into a fresh tmpfs
file and a parallel memcpy into a shared mapping of one."""
import mmap, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frb_baseband_amd import channeliser as ch, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
raw = synth.make_vdif(secs, bw_mhz=32.0, nchan=1024)
vd, fil, fifo = "/dev/shm/frbch_t.vdif", "/dev/shm/frbch_t.fil", "/dev/shm/frbch_t.fifo"
raw.tofile(vd)
nbytes = raw.size
def best(f, n=4):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)
# OS ceilings
blob = raw.tobytes()
def os_write():
    if os.path.exists(fil): os.remove(fil)
    with open(fil, "wb", buffering=0) as f: f.write(blob)
t = best(os_write); print("OS: one write() stream into a fresh tmpfs file: %.1f ms = %.2f GB/s" % (t * 1e3, nbytes / t / 1e9))
def os_read():
    with open(vd, "rb", buffering=0) as f: f.readinto(bytearray(nbytes))
t = best(os_read); print("OS: one read() stream from tmpfs: %.1f ms = %.2f GB/s" % (t * 1e3, nbytes / t / 1e9))
def os_mmap(nthr=8):
    if os.path.exists(fil): os.remove(fil)
    fd = os.open(fil, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644); os.ftruncate(fd, nbytes)
    m = np.memmap(fil, dtype=np.uint8, mode="r+", shape=(nbytes,))
    per = (nbytes + nthr - 1) // nthr
    th = [threading.Thread(target=lambda i=i: m.__setitem__(slice(i * per, min(nbytes, (i + 1) * per)), raw[i * per:min(nbytes, (i + 1) * per)])) for i in range(nthr)]
    [x.start() for x in th]; [x.join() for x in th]; del m; os.close(fd)
t = best(os_mmap); print("OS: 8 threads copying into a shared mapping of a fresh tmpfs file: %.1f ms = %.2f GB/s" % (t * 1e3, nbytes / t / 1e9))
cfg = ch.new_config(bw_mhz=32.0, nchan=1024, total_s=secs, rescale_constant=1)
with ch.Channeliser(cfg) as c:
    def to_file():
        c.reset()
        if os.path.exists(fil): os.remove(fil)
        c.run_file(vd, fil)
    to_file()
    t = best(to_file); print("frbch_run_file -> tmpfs file: %.1f ms = %.1f Gsamples/s" % (t * 1e3, nbytes * 2 / t / 1e9 * 8000 / 8032))
    def to_null():
        c.reset(); c.run_file(vd, "/dev/null")
    t = best(to_null); print("frbch_run_file -> /dev/null:   %.1f ms = %.1f Gsamples/s" % (t * 1e3, nbytes * 2 / t / 1e9 * 8000 / 8032))
    if os.path.exists(fifo): os.remove(fifo)
    os.mkfifo(fifo)
    def to_fifo():
        c.reset()
        def drain():
            with open(fifo, "rb", buffering=0) as f:
                buf = bytearray(1 << 22)
                while f.readinto(buf): pass
        th = threading.Thread(target=drain); th.start(); c.run_file(vd, fifo); th.join()
    t = best(to_fifo, 2); print("frbch_run_file -> FIFO (drained): %.1f ms = %.1f Gsamples/s" % (t * 1e3, nbytes * 2 / t / 1e9 * 8000 / 8032))
for f in (vd, fil, fifo):
    if os.path.exists(f): os.remove(f)
