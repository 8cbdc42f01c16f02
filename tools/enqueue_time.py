"""Diagnostic: host time to QUEUE one step of a bench workload (the call returns) against the step's device time, with the per-kernel
timing events on and off.  usage: enqueue_time.py [bench.py arguments]   (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
args = bench.build_parser().parse_args(sys.argv[1:])
spec = bench.workload_spec(args)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload(torch, dev, spec, args, 0, 0)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
for prof in (False, True, False):
    for c in wl.chans:
        c.set_profiling(prof)
        c.timing_reset()
    host, total = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    print(f"overlap {args.overlap} profiling {prof}: host enqueue ms {['%.2f' % x for x in host]}  step ms {['%.2f' % x for x in total]}", flush=True)
wl.close()
