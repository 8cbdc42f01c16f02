"""Digest of the K1 phase stamps (FRBCH_STAMPS=<file> python bench.py ...): median cycles per phase over waves."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = a[(a[:, 0] > 0) & (a[:, 10] > 0)]
names = ["unpack", "fwd passes", "delay", "bwd passes", "barrier 1", "refill", "sweep 0", "sweep 1", "(tail)", "barrier 2"]
d = np.diff(a[:, :11], axis=1)
print("waves with stamps:", len(a), " iteration (stamp 0 -> 10): median %d cycles" % np.median(a[:, 10] - a[:, 0]))
for i, n in enumerate(names):
    print("%-11s median %6d  p10 %6d  p90 %6d" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
w = a.reshape(-1, nw, 16) if len(a) % nw == 0 else None
if w is not None:
    skew = (w[:, :, 4].max(axis=1) - w[:, :, 4].min(axis=1))
    print("arrival skew at barrier 1 within a workgroup: median %d  p90 %d" % (np.median(skew), np.percentile(skew, 90)))
    print("start skew (stamp 0) within a workgroup: median %d" % np.median(w[:, :, 0].max(axis=1) - w[:, :, 0].min(axis=1)))
