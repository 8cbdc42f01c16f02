#!/usr/bin/env python3
"""Benchmark of the VDIF -> SIGPROC-filterbank hot path on MI355X.

One "step" = one pass of the whole hot path (frame parse + 2-bit unpack -> filterbank -> detect ->
tscrunch -> rescale -> 8-bit digitise) over one batch of synthetic per-IF VDIF that is already
resident in HBM, through the C ABI (include/frbch.h).  Workload at N=1 is BASELINE.json
configs[1]: 1 IF, 32 MHz, 2-bit dual-pol -> 1024-channel Stokes-I (digifil flags
`-c -b8 -d1 -F1024:2048`, process_vdif.py:157-171).  With N GPUs every rank owns one IF (the
path shards by IF, base2fil.sh:60-66: no collective on the data path) -> weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":       dominant kernel's algorithmic bytes / its HIP-event time vs the 8 TB/s HBM peak; "traffic" = HBM bytes
                    per launch of that kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, gfx950
                    x2 correction of wide fetches) collected LIVE by child processes of this very run (N = 1), else null
  "cpu_baseline":   the CPU oracle port timed on a bounded sample of the same workload (rank 0, N=1)
  "host_inclusive": the product call the reference makes (frbch_run_file: VDIF file -> .fil file, both on tmpfs) for the
                    same workload, PCIe both ways included -- never `value`

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process only LAUNCHES N fresh rank processes (before any
GPU call; it never touches the GPU itself), forwards rank 0's JSON line and exits non-zero if a rank failed.  Under
torch.distributed.run (the driver's way) WORLD_SIZE is set and the process is a rank.
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
# fp32 vector peak in lane operations: 157.3 TFLOP/s (spec, FMA = 2) / 2 = 78.6 T lane-ops/s; tools_micro/valu_peak measures
# 71 - 75 T/s for v_add / v_sub streams at 16 waves per CU, 63 at 8 (a packed v_pk_* counts 2 and issues at half the rate)
VALU_PEAK_TLOPS = 78.65
# static VALU instruction count per wave and filterbank block of the dominant kernels (llvm-objdump of the loop body;
# SQ_INSTS_VALU of tools_pmc.sh agrees): waves per workgroup x workgroups per block follow from the geometry
VALU_PER_WAVE_BLOCK = {"frbch_k1_wave<3,8,1>": (2424, 8)}


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


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps; the first steps on a fresh box run ~5 %% slower (clock / power state settle)")
    ap.add_argument("--seconds", type=float, default=10.0, help="seconds of one IF per step (SURVEY 8d: 10 s)")
    ap.add_argument("--nchan", type=int, default=1024)
    ap.add_argument("--bw", type=float, default=32.0)
    ap.add_argument("--pol", type=int, default=2)
    ap.add_argument("--tscrunch", type=int, default=1)
    ap.add_argument("--dm", type=float, default=0.0)
    ap.add_argument("--coherent", action="store_true", help="-F C:D: coherent dedispersion inside the filterbank (cfg 5)")
    ap.add_argument("--freq", type=float, default=1608.0, help="centre sky frequency, MHz (matters with --coherent)")
    ap.add_argument("--freq-res", type=int, default=0)
    ap.add_argument("--maxb", type=int, default=0, help="filterbank blocks per kernel launch (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the host-inclusive (frbch_run_file) leg")
    ap.add_argument("--host-runs", type=int, default=3, help="timed frbch_run_file passes of the host-inclusive leg")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live PMC passes (roofline.traffic = null)")
    ap.add_argument("--flags", type=int, default=0, help="kernel-selection flags of frbch_config (see include/frbch.h)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 (gloo)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # run under rocprofv3 by collect_traffic
    return ap


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N without WORLD_SIZE: start N fresh rank processes (one per GPU, like one digifil per IF,
    base2fil.sh:60-66).  This parent never initialises the GPU; a failing rank fails the run (no re-exec, no retry)."""
    import subprocess
    n = args.gpus
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if args.share_gpu else r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = procs[0].communicate()[0]
    rcs = [procs[0].returncode] + [pr.wait() for pr in procs[1:]]
    if any(rcs):
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        sys.stdout.write(out0 or "")
        return 1
    # ONE JSON line: whatever else a rank printed on stdout (gloo's connection banner in the one-GPU rehearsal) goes to stderr
    lines = (out0 or "").splitlines()
    keep = [ln for ln in lines if ln.startswith('{"metric"')]
    for ln in lines:
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if not keep:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(keep[-1], flush=True)
    return 0


WIDE_FETCH = ("frbch_k2_", "frbch_quantise", "frbch_stats_partial", "frbch_k0_stage", "frbch_k1_wave", "frbch_k1_split")


def short_kernel_name(profiler_name):
    """'void fast::frbch_k1_wave<3, 8, 1, true, false>(KParams)' -> 'frbch_k1_wave<3,8,1>': the name of the engine's timing
    slot (frbch_get_timing), which carries no staging / coherent / statistics flags; None for other kernels"""
    import re
    m = re.search(r"(frbch_[a-z0-9_]+)(<[^>]*>)?", profiler_name)
    if not m:
        return None
    short = (m.group(1) + (m.group(2) or "")).replace(" ", "")
    return re.sub(r"(,(true|false))+>", ">", short)


def collect_traffic(argv):
    """HBM bytes per launch of every frbch kernel, measured NOW: two child runs of this script under rocprofv3
    (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`: the counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
    FETCH_SIZE is doubled for kernels that read with 16-byte-per-lane loads (gfx950 tallies 128-B requests at 64 B, same
    guide, "HBM").  The caller has not touched the GPU yet.  Any failure -> None (the bench line then says traffic null)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    child = [a for a in argv if a not in ("--pmc-child",)]
    res = {}
    tmp = tempfile.mkdtemp(prefix="frbch_pmc_", dir="/tmp")
    try:
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, cnt)
            cmd = ["rocprofv3", "--pmc", cnt, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__)] + child + ["--pmc-child"]
            env = dict(os.environ, TMPDIR="/tmp")
            pr = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
            if pr.returncode != 0:
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None
            for path in files:
                for row in csv.DictReader(open(path)):
                    if row.get("Counter_Name") != cnt or "frbch" not in row.get("Kernel_Name", ""):
                        continue
                    k = row["Kernel_Name"]
                    rec = res.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {}})
                    rec[cnt] += float(row["Counter_Value"])
                    rec["n"][cnt] = rec["n"].get(cnt, 0) + 1
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"_source": "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (KB counters; wide fetches x2)"}
    for k, rec in res.items():
        short = short_kernel_name(k)
        if not short:
            continue
        nl = max(1, min(rec["n"].get("FETCH_SIZE", 1), rec["n"].get("WRITE_SIZE", 1)))
        corr = 2.0 if short.startswith(WIDE_FETCH) else 1.0
        out[short] = (rec["FETCH_SIZE"] / max(1, rec["n"].get("FETCH_SIZE", 1)) * corr +
                      rec["WRITE_SIZE"] / max(1, rec["n"].get("WRITE_SIZE", 1))) * 1024.0
        out.setdefault("_launches", {})[short] = nl
    return out


PCIE_PEAK_GBS = 64.0   # PCIe Gen5 x16, one direction (MI355X_MICROARCH.md host link)


def host_inclusive(args, torch, dist, ch, cfg_kwargs, frames, nfr, rank, world, samples_per_step):
    """What the reference's call does end to end (process_vdif.py:191 `digifil ... -o out hdr`): frbch_run_file from a
    VDIF file to a .fil file, both on tmpfs -- disk excluded, PCIe both ways and the host threads included."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    vd = os.path.join(base, f"frbch_bench_{os.getpid()}_if{rank}.vdif")
    fil = vd.replace(".vdif", ".fil")
    try:
        frames[: nfr * 8032].cpu().numpy().tofile(vd)
        in_bytes = os.path.getsize(vd)
        times = []
        with ch.Channeliser(ch.new_config(**cfg_kwargs)) as c:
            for i in range(args.host_runs + 1):          # first pass untimed: pinned buffers, page cache, clocks
                c.reset()
                if os.path.exists(fil):
                    os.remove(fil)        # what run_digifil does with --force before the call (process_vdif.py:146-149)
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                c.run_file(vd, fil)
                dt = time.perf_counter() - t0
                if dist is not None:
                    t = torch.tensor([dt], dtype=torch.float64, device=frames.device if args.backend == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                if i:
                    times.append(dt)
        out_bytes = os.path.getsize(fil)
        best = sorted(times)[len(times) // 2]
        each_way = max(in_bytes, out_bytes) / best / 1e9
        return {"value": round(samples_per_step * world / best / 1e6, 1), "unit": "Msamples/s", "s_per_scan": round(best, 4),
                "realtime_x": round(samples_per_step / best / 1e6 / (2 * args.bw), 1),
                "bytes_in_per_if": in_bytes, "bytes_out_per_if": out_bytes,
                "pcie": {"bound": "pcie", "achieved": round(each_way, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s per direction per GPU",
                         "frac": round(each_way / PCIE_PEAK_GBS, 4)},
                "path": f"frbch_run_file: {base} VDIF -> pinned ring -> HBM -> pinned ring -> {base} .fil, median of {len(times)} scans after one untimed"}
    except Exception as exc:   # a reported extra: never fail the bench for it
        return {"error": repr(exc)}
    finally:
        for f in (vd, fil):
            try:
                os.remove(f)
            except OSError:
                pass


def main():
    args = build_parser().parse_args()
    argv = sys.argv[1:]
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch_ranks(args, argv))          # parent: launches only, never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.backend is None:
        args.backend = "gloo" if args.share_gpu else "nccl"

    # live PMC passes first: the children must start before this process initialises the GPU
    live_traffic = None
    if world == 1 and not args.no_traffic and not args.pmc_child:
        live_traffic = collect_traffic(argv + ["--steps", "1", "--warmup", "1", "--no-cpu", "--no-host", "--no-traffic"])
    if args.pmc_child:
        args.steps, args.warmup, args.no_cpu, args.no_host = 1, 1, True, True

    import torch
    if world > 1:
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
    cfg_kwargs = dict(bw_mhz=args.bw, nchan=args.nchan, pol_mode=args.pol, nbit_out=8, tscrunch=args.tscrunch,
                      rescale_constant=1, rescale_interval_s=10.0, total_s=args.seconds, device=local_rank,
                      max_blocks_per_launch=args.maxb, flags=args.flags, dm=args.dm,
                      coherent=1 if args.coherent else 0, freq_mhz=args.freq, freq_res=args.freq_res)
    cfg = ch.new_config(**cfg_kwargs)
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
        c.reset()     # (waits for the previous step's work on `stream` before it touches the rescale state)
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
    dt_rank = dt
    per_rank = None
    if dist is not None:
        red_dev = dev if args.backend == "nccl" else "cpu"
        tall = [torch.zeros(1, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], device=red_dev, dtype=torch.float64))
        per_rank = [float(t.item()) for t in tall]
        dt = max(per_rank)

    timing = c.get_timing()
    if args.pmc_child:
        c.close()
        return
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
    c.close()
    del out
    host = None
    if not args.no_host:
        host = host_inclusive(args, torch, dist, ch, cfg_kwargs, frames, nfr, rank, world, samples_per_step)
    if rank == 0:
        total_samples = samples_per_step * args.steps * world
        value = total_samples / dt / 1e6
        dom = max(timing.items(), key=lambda kv: kv[1]["total_ms"])
        name, rec = dom
        ach = rec["algorithmic_bytes"] / (rec["total_ms"] * 1e-3) / 1e9 if rec["total_ms"] > 0 else 0.0
        # measured HBM bytes per launch of that kernel: live PMC passes of this run (collect_traffic), never a stored constant
        traffic = None
        if live_traffic and name in live_traffic:
            traffic = live_traffic[name]
        valu = None
        if name in VALU_PER_WAVE_BLOCK and rec["total_ms"] > 0:     # the other roof of this kernel: fp32 VALU lane operations
            per_wave, waves = VALU_PER_WAVE_BLOCK[name]
            lane_ops = per_wave * waves * 64.0 * (2 * args.nchan // 8) * nblocks * rec["launches"]
            tl = lane_ops / (rec["total_ms"] * 1e-3) / 1e12
            valu = {"bound": "valu", "achieved": round(tl, 2), "peak": VALU_PEAK_TLOPS, "unit": "T lane-ops/s (fp32, unpacked)",
                    "frac": round(tl / VALU_PEAK_TLOPS, 4), "instructions_per_wave_and_block": per_wave}
        roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": (live_traffic or {}).get("_source"),
                "avg_launch_ms": round(rec["total_ms"] / max(1, rec["launches"]), 5),
                "algorithmic_bytes_per_launch": rec["algorithmic_bytes"] / max(1, rec["launches"]),
                "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in timing.items()},
                "valu": valu,
                "whole_path": {"algorithmic_bytes_per_sample": 17.0 if args.pol < 4 else 18.5,
                               "achieved": round(value / world * 1e6 * (17.0 if args.pol < 4 else 18.5) / 1e9, 1),
                               "frac": round(value / world * 1e6 * (17.0 if args.pol < 4 else 18.5) / 1e9 / HBM_PEAK_GBS, 4)}}
        prod = {2: "Stokes-I", 4: "coherency (-d4)", 5: "IQUV"}.get(args.pol, "pol%d" % args.pol)
        dflag = {2: "-d1", 3: "-d3", 4: "-d4", 5: "-d4 -iquv"}.get(args.pol, "-P%d" % args.pol)
        line = {
            "metric": "Msamples/s channelised to .fil per GPU; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{world} IF x {args.bw:g} MHz 2-bit dual-pol VDIF -> {args.nchan}-ch "
                                   f"{prod} 8-bit .fil "
                                   f"(-c -b8 {dflag}{' -t %d' % args.tscrunch if args.tscrunch > 1 else ''} -F{args.nchan}:{info.freq_res}"
                                   f"{' -D %g -F%d:D' % (args.dm, args.nchan) if args.coherent else ''}), {args.seconds:g} s per IF per step, "
                                   f"one IF per GPU (no data-path collective), first rescale interval measured every step, frames and .fil rows resident in HBM",
                       "samples_per_step_per_gpu": samples_per_step, "blocks_per_step": nblocks,
                       "realtime_x": round(value / world / (2 * args.bw), 2),
                       "steady_state_msamples_per_gpu": round(steady, 1),
                       "per_rank_seconds": per_rank},
            "roofline": roof,
            "host_inclusive": host,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.bw, args.nchan)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
