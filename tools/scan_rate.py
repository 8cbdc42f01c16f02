"""Host-inclusive rate of frbch_run_scan: nif IFs x secs of synthetic VDIF in tmpfs -> one IFall file (not `value`)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frb_baseband_amd import multi_if, synth
nif, secs, pol = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
d = tempfile.mkdtemp(dir="/dev/shm")
raw = synth.make_vdif(1.0, bw_mhz=32.0, nchan=1024)
vd = {}
for i in range(1, nif + 1):
    vd[i] = os.path.join(d, f"x_ef_no0001_IF{i}.vdif")
    with open(vd[i], "wb") as f:
        for _ in range(int(secs)):
            raw.tofile(f)
for direct in (True, False):
    t0 = time.perf_counter()
    out = multi_if.process_scan(vd, freq_lsb_0=1340.0, bw=32.0, nchan=1024, nsec=secs, out_dir=d, direct=direct, pol=pol,
                                ra="01:00:00", dec="02:00:00")
    dt = time.perf_counter() - t0
    print(f"{'run_scan (device concat)' if direct else 'per-IF files + host splice'}: {nif} IFs x {secs:g} s, pol {pol}: "
          f"{dt:.2f} s = {nif * secs * 64e6 / dt / 1e9:.2f} Gsamples/s = {secs / dt:.1f} x real time, output {os.path.getsize(out) / 1e6:.0f} MB", flush=True)
    for f in os.listdir(d):
        if f.endswith(".fil") or f.endswith(".hdr"):
            os.remove(os.path.join(d, f))
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
