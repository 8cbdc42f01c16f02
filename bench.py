#!/usr/bin/env python3
"""Benchmark of the VDIF -> SIGPROC-filterbank hot path on MI355X.

One "step" = one pass of the whole hot path (frame parse + 2-bit unpack -> filterbank -> detect ->
tscrunch -> rescale -> 8-bit digitise) over one batch of synthetic per-IF VDIF that is already
resident in HBM, through the C ABI (include/frbch.h).  Workload at N=1 is BASELINE.json
configs[1]: 1 IF, 32 MHz, 2-bit dual-pol -> 1024-channel Stokes-I (digifil flags
`-c -b8 -d1 -F1024:2048`, process_vdif.py:157-171).  With N GPUs every rank owns one IF (the
path shards by IF, base2fil.sh:60-66: no collective on the data path) -> weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     dominant kernel's algorithmic bytes / its HIP-event time vs the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle port timed on a bounded sample of the same workload (rank 0, N=1)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)


def synth_frames_device(torch, dev, seconds: float, bw_mhz: float, nchan: int, if_index: int, payload: int = 8000):
    """Seeded synthetic per-IF VDIF built on the GPU (same recipe as frb_baseband_amd.synth:
    unit Gaussian noise per pol + weak tone at channel nchan//3, 2-bit at +-0.9816 sigma,
    8032-byte frames).  Returns (uint8 tensor of whole frames, nframes)."""
    import math
    fps = int(round(abs(bw_mhz) * 1.0e6 * 2 * 2 * 2 / 8 / payload))
    nfr = int(round(seconds * fps))
    spf = payload * 2
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xF4B0 + if_index)
    frames = torch.zeros((nfr, 32 + payload), dtype=torch.uint8, device=dev)
    chunk_fr = max(1, (1 << 24) // spf)
    k = nchan // 3
    w = 2.0 * math.pi * ((k + 0.5) / (2.0 * nchan))
    for f0 in range(0, nfr, chunk_fr):
        nf = min(chunk_fr, nfr - f0)
        n = nf * spf
        x = torch.randn((2, n), generator=gen, device=dev, dtype=torch.float32)
        t = torch.arange(f0 * spf, f0 * spf + n, device=dev, dtype=torch.float64)
        ph = (w * t) % (2.0 * math.pi)
        x[0] += (0.1 * math.sqrt(2.0)) * torch.cos(ph).float()
        x[1] += (0.1 * math.sqrt(2.0)) * torch.sin(ph).float()
        st = ((x >= -0.9816).to(torch.uint8) + (x >= 0).to(torch.uint8) + (x >= 0.9816).to(torch.uint8))
        b = st[0, 0::2] | (st[1, 0::2] << 2) | (st[0, 1::2] << 4) | (st[1, 1::2] << 6)
        frames[f0:f0 + nf, 32:] = b.view(nf, payload)
        del x, t, ph, st, b
    # headers: word0 seconds, word1 frame# | epoch, word2 frame length/8 | log2 nchan, word3 bits-1
    idx = torch.arange(nfr, device=dev, dtype=torch.int64)
    words = torch.zeros((nfr, 8), dtype=torch.int32, device=dev)
    words[:, 0] = (1000 + idx // fps).to(torch.int32)
    words[:, 1] = ((40 << 24) | (idx % fps)).to(torch.int32)
    words[:, 2] = (1 << 24) | ((32 + payload) // 8)
    words[:, 3] = (1 << 26) | 0x4566
    frames[:, :32] = words.view(torch.uint8).view(nfr, 32)
    return frames.reshape(-1), nfr


def cpu_baseline(seconds_budget: float, bw: float, nchan: int):
    """CPU restatement (oracle) timed on a bounded sample of the same workload; reported, not the target."""
    import numpy as np
    from frb_baseband_amd import synth
    from oracle import frb_oracle as o
    try:
        from oracle import c_oracle
    except Exception:
        c_oracle = None
    r = o.freq_res_for(nchan)
    n = 2 * nchan * r
    raw = synth.make_vdif(2 * n / (2e6 * bw) + 0.001, bw_mhz=bw, nchan=nchan)
    if c_oracle is not None and c_oracle.available():
        t0 = time.perf_counter()
        nblk = 0
        while time.perf_counter() - t0 < seconds_budget:
            c_oracle.channelise_blocks(raw, bw, nchan, r, 2)
            nblk += 2
        dt = time.perf_counter() - t0
        res = {"value": nblk * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
               "sample": f"{nblk} filterbank blocks of {n} dual-pol samples, C fp32 port (oracle/frb_oracle.c), 1 thread"}
        # all host cores, the way the reference fans out: one single-threaded process per IF (base2fil.sh:60-66,219).
        # Child processes never touch the GPU.
        try:
            import subprocess
            # the GPU box gives a one-GPU job a share of 16 host cores: do not take more
            ncpu = min(len(os.sched_getaffinity(0)), int(os.environ.get("FRBCH_CPU_WORKERS", "16")))
            procs = [subprocess.Popen([sys.executable, "-m", "oracle.c_oracle", str(seconds_budget / 2), str(bw), str(nchan), str(i)],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                     for i in range(ncpu)]
            rates = [float(pr.communicate(timeout=seconds_budget * 4 + 120)[0].strip().splitlines()[-1]) for pr in procs]
            res["all_cores"] = {"value": sum(rates) / 1e6, "unit": "Msamples/s", "cores": ncpu,
                                "sample": f"{ncpu} single-threaded processes, one IF each, {seconds_budget / 2:g} s"}
        except Exception as exc:   # reported baseline only: never fail the bench for it
            res["all_cores"] = {"error": str(exc)}
        return res
    cfg = o.Config(bw_mhz=bw, nchan=nchan, total_s=10.0)
    t0 = time.perf_counter()
    nblk = 0
    while time.perf_counter() - t0 < seconds_budget:
        o.channelise(raw, cfg)
        nblk += 2
    dt = time.perf_counter() - t0
    return {"value": nblk * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} filterbank blocks of {n} dual-pol samples, numpy fp64 oracle (pocketfft), 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps; the first steps on a fresh box run ~5 %% slower (clock / power state settle)")
    ap.add_argument("--seconds", type=float, default=10.0, help="seconds of one IF per step (SURVEY 8d: 10 s)")
    ap.add_argument("--nchan", type=int, default=1024)
    ap.add_argument("--bw", type=float, default=32.0)
    ap.add_argument("--pol", type=int, default=2)
    ap.add_argument("--dm", type=float, default=0.0)
    ap.add_argument("--coherent", action="store_true", help="-F C:D: coherent dedispersion inside the filterbank (cfg 5)")
    ap.add_argument("--freq", type=float, default=1608.0, help="centre sky frequency, MHz (matters with --coherent)")
    ap.add_argument("--freq-res", type=int, default=0)
    ap.add_argument("--maxb", type=int, default=0, help="filterbank blocks per kernel launch (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="debug flags of frbch_config (see include/frbch.h)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        dist = None
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU fallback of the hot path)", file=sys.stderr)
        sys.exit(2)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from frb_baseband_amd import channeliser as ch
    cfg = ch.new_config(bw_mhz=args.bw, nchan=args.nchan, pol_mode=args.pol, nbit_out=8, tscrunch=1,
                        rescale_constant=1, rescale_interval_s=10.0, total_s=args.seconds, device=local_rank,
                        max_blocks_per_launch=args.maxb, flags=args.flags, dm=args.dm,
                        coherent=1 if args.coherent else 0, freq_mhz=args.freq, freq_res=args.freq_res)
    c = ch.Channeliser(cfg)
    info = c.info
    frames, nfr = synth_frames_device(torch, dev, args.seconds, args.bw, args.nchan, if_index=rank)
    torch.cuda.synchronize()   # the library runs on its own stream: inputs must be complete
    nblocks = (nfr * 8000 - info.block_payload_bytes) // info.block_stride_bytes + 1   # overlap-save when --coherent
    rows = nblocks * info.rows_per_block
    out = torch.empty(rows * info.row_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    samples_per_step = nblocks * info.block_stride_bytes * 2      # new dual-pol samples consumed (2-bit: 2 per byte)

    def step():
        c.reset()
        r1 = c.process_device(frames.data_ptr(), nfr, 8032, 32, 0, nblocks, out.data_ptr(), out.numel(), stream)
        r2 = c.flush_device(out.data_ptr() + r1 * info.row_bytes, out.numel() - r1 * info.row_bytes, stream)
        return r1 + r2

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    c.set_profiling(True)
    c.timing_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got_rows = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert got_rows == rows, (got_rows, rows)
    if dist is not None:
        tmax = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    timing = c.get_timing()
    # extra (not `value`): steady state of a long scan -- scale frozen after the first interval (-c), K2 digitises in-kernel
    c.set_profiling(False)
    off, sc = c.get_rescale()
    c.reset()
    c.set_rescale(off, sc)
    c.process_device(frames.data_ptr(), nfr, 8032, 32, 0, nblocks, out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        c.process_device(frames.data_ptr(), nfr, 8032, 32, 0, nblocks, out.data_ptr(), out.numel(), stream)
    torch.cuda.synchronize()
    steady = samples_per_step * args.steps / (time.perf_counter() - t1) / 1e6
    if rank == 0:
        total_samples = samples_per_step * args.steps * world
        value = total_samples / dt / 1e6
        dom = max(timing.items(), key=lambda kv: kv[1]["total_ms"])
        name, rec = dom
        ach = rec["algorithmic_bytes"] / (rec["total_ms"] * 1e-3) / 1e9 if rec["total_ms"] > 0 else 0.0
        # measured HBM bytes per launch of that kernel: PMC passes (FETCH_SIZE / WRITE_SIZE, collected
        # separately with rocprofv3 by tools_profile.sh) stored per filterbank block in profiles/hbm_traffic.json
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            rec_t = tj["kernels"].get(name)
            if rec_t and args.nchan == 1024 and args.bw == 32.0 and args.pol == 2 and args.flags == 0 and not args.coherent:
                per_block = (rec_t["fetch_kb_per_block"] * rec_t["fetch_correction"] + rec_t["write_kb_per_block"]) * 1024.0
                traffic = per_block * nblocks * args.steps / max(1, rec["launches"])
        except Exception:
            traffic = None
        roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(rec["total_ms"] / max(1, rec["launches"]), 5),
                "algorithmic_bytes_per_launch": rec["algorithmic_bytes"] / max(1, rec["launches"]),
                "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in timing.items()}}
        line = {
            "metric": "Msamples/s channelised to .fil per GPU; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{world} IF x {args.bw:g} MHz 2-bit dual-pol VDIF -> {args.nchan}-ch "
                                   f"{'Stokes-I' if args.pol == 2 else 'pol%d' % args.pol} 8-bit .fil "
                                   f"(-c -b8 -d1 -F{args.nchan}:{info.freq_res}"
                                   f"{' -D %g -F%d:D' % (args.dm, args.nchan) if args.coherent else ''}), {args.seconds:g} s per IF per step, "
                                   f"one IF per GPU, first rescale interval measured every step",
                       "samples_per_step_per_gpu": samples_per_step, "blocks_per_step": nblocks,
                       "realtime_x": round(value / world / (2 * args.bw), 2),
                       "steady_state_msamples_per_gpu": round(steady, 1)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.bw, args.nchan)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
