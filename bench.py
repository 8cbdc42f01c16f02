#!/usr/bin/env python3
"""Benchmark of the VDIF -> SIGPROC-filterbank hot path on MI355X.

One "step" = one pass of the whole hot path (frame parse + 2-bit unpack -> filterbank -> detect ->
tscrunch -> rescale -> 8-bit digitise) over one batch of synthetic per-IF VDIF that is already
resident in HBM, through the C ABI (include/frbch.h).  Workload at N=1 is BASELINE.json
configs[2], the largest single-GPU configuration: 8 IFs x 32 MHz, 2-bit dual-pol -> 1024-channel
full-Stokes IQUV on one GPU (per IF the digifil flags `-c -b8 -d4 -F1024:2048`,
process_vdif.py:157-171), all IFs into ONE row buffer through frbch_scan_device.  With N GPUs every
rank owns its own 8 IFs (the path shards by IF, base2fil.sh:60-66: no collective on the data path)
-> weak scaling.  The other BASELINE configurations ride in the same line under "configs".

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":       the SURVEY 8(d) figure.  `achieved` / `frac` = WHOLE PATH: value x the 8(d) budget bytes per sample
                    (0.502 in + 8 + 8 spill + output codes; 18.5 B for four 8-bit products) against the 8 TB/s HBM peak --
                    reproducible from the driver's own clock as budget x samples_per_step / ms_per_step.  `kernel*` = the
                    kernel with the largest summed launch time (no exclusions), priced with ITS share of that budget (K1
                    0.502 + 8, K2 8 + output codes, digitiser / statistics / K0 nothing) over its HIP-event time.
                    `traffic` = HBM bytes of ALL kernels of one step from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE:
                    separate passes, gfx950 x2 correction of wide fetches) collected LIVE by child processes of this very
                    run (N = 1), else null; `traffic_ratio` = that over budget x samples_per_step.  The per-kernel byte
                    model of the engine (which counts the fp32 rows of a buffered rescale interval as a kernel's own
                    bytes) is kept apart as `kernel_model_frac` / `per_kernel`.
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


def cpu_baseline(seconds_budget: float, bw: float, nchan: int, pol: int = 2):
    """CPU restatement (oracle) timed on a bounded sample of the same workload (same products); reported, not the target."""
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
            c_oracle.channelise_blocks(raw, bw, nchan, r, 2, pol)
            nblk += 2
        dt = time.perf_counter() - t0
        res = {"value": nblk * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
               "sample": f"{nblk} filterbank blocks of {n} dual-pol samples of one IF, pol_mode {pol}, C fp32 port (oracle/frb_oracle.c), 1 thread"}
        # all host cores, the way the reference fans out: one single-threaded process per IF (base2fil.sh:60-66,219).
        # Child processes never touch the GPU.
        try:
            import subprocess
            # the GPU box gives a one-GPU job a share of 16 host cores: do not take more
            ncpu = min(len(os.sched_getaffinity(0)), int(os.environ.get("FRBCH_CPU_WORKERS", "16")))
            procs = [subprocess.Popen([sys.executable, "-m", "oracle.c_oracle", str(seconds_budget / 2), str(bw), str(nchan), str(i), str(pol)],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                     for i in range(ncpu)]
            rates = [float(pr.communicate(timeout=seconds_budget * 4 + 120)[0].strip().splitlines()[-1]) for pr in procs]
            res["all_cores"] = {"value": sum(rates) / 1e6, "unit": "Msamples/s", "cores": ncpu,
                                "sample": f"{ncpu} single-threaded processes, one IF each, {seconds_budget / 2:g} s"}
            res["all_cores_value"] = sum(rates) / 1e6       # (flat copies: BASELINE.md section 3 asks for both figures)
            res["all_cores_n"] = ncpu
        except Exception as exc:   # reported baseline only: never fail the bench for it
            res["all_cores"] = {"error": str(exc)}
        return res
    cfg = o.Config(bw_mhz=bw, nchan=nchan, total_s=10.0, pol_mode=pol)
    t0 = time.perf_counter()
    nblk = 0
    while time.perf_counter() - t0 < seconds_budget:
        o.channelise(raw, cfg)
        nblk += 2
    dt = time.perf_counter() - t0
    return {"value": nblk * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} filterbank blocks of {n} dual-pol samples, numpy fp64 oracle (pocketfft), 1 thread"}


# BASELINE.json configurations as workloads of this bench (per GPU).  `nif` IFs share the GPU: their rows land in ONE row buffer
# [row][product][IF-major channels] through frbch_scan_device (the frequency concatenation of base2fil.sh:422 in the store addresses)
WORKLOADS = {
    # configs[2]: the largest single-GPU configuration = the default (golden argv process_vdif.py:166-171 `-d4 -F1024:2048`, IQUV
    # formed from the four products as north_star says)
    "cfg3": dict(nif=8, bw=32.0, nchan=1024, pol=5, tscrunch=1, seconds=10.0),
    "cfg2": dict(nif=1, bw=32.0, nchan=1024, pol=2, tscrunch=1, seconds=10.0),                 # configs[1]
    "cfg4": dict(nif=2, bw=64.0, nchan=4096, pol=2, tscrunch=8, seconds=10.0),                 # configs[3]: one GPU's share (2 IF)
    "cfg5": dict(nif=1, bw=32.0, nchan=2048, pol=2, tscrunch=1, seconds=10.0, freq_res=4096, dm=56.7, coherent=True, freq=1400.0),
    "cfg1": dict(nif=1, bw=16.0, nchan=128, pol=2, tscrunch=1, seconds=10.0),                  # configs[0] shape (parity case)
    # what the online chain cuts an IF into (submit_job.py:74-105, MinChanPerIF = 32): 32 channels per 32 MHz IF, freq_res 512
    "online32": dict(nif=1, bw=32.0, nchan=32, pol=2, tscrunch=1, seconds=10.0),
}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps; the first steps on a fresh box run ~5 %% slower (clock / power state settle)")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS), help="BASELINE configuration (default: configs[2])")
    ap.add_argument("--nif", type=int, default=None, help="IFs per GPU (one handle each, one row buffer)")
    ap.add_argument("--seconds", type=float, default=None, help="seconds of every IF per step (SURVEY 8d: 10 s)")
    ap.add_argument("--nchan", type=int, default=None)
    ap.add_argument("--bw", type=float, default=None)
    ap.add_argument("--pol", type=int, default=None)
    ap.add_argument("--tscrunch", type=int, default=None)
    ap.add_argument("--dm", type=float, default=None)
    ap.add_argument("--coherent", action="store_true", default=None, help="-F C:D: coherent dedispersion inside the filterbank (cfg 5)")
    ap.add_argument("--freq", type=float, default=None, help="centre sky frequency, MHz (matters with --coherent)")
    ap.add_argument("--freq-res", type=int, default=None)
    ap.add_argument("--maxb", type=int, default=0, help="filterbank blocks per kernel launch (0 = library default)")
    ap.add_argument("--overlap", type=int, default=0, help="frbch_config.overlap: front-lane CUs | batches << 16 (0 = automatic, 1 = no overlap)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the host-inclusive (frbch_run_file / frbch_run_scan) leg")
    ap.add_argument("--host-runs", type=int, default=3, help="timed passes of the host-inclusive leg")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live PMC passes (roofline.traffic = null)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state extra (profiler runs: only the timed workload's launches)")
    ap.add_argument("--config-steps", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0, help="kernel-selection flags of frbch_config (see include/frbch.h)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 (gloo)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # run under rocprofv3 by collect_traffic
    return ap


def workload_spec(args, name=None):
    """the workload's parameters, command-line values on top of the named configuration"""
    spec = dict(nif=1, bw=32.0, nchan=1024, pol=2, tscrunch=1, seconds=10.0, freq_res=0, dm=0.0, coherent=False, freq=1608.0)
    spec.update(WORKLOADS[name or args.workload])
    if name is None:
        for key, arg in (("nif", args.nif), ("bw", args.bw), ("nchan", args.nchan), ("pol", args.pol), ("tscrunch", args.tscrunch),
                         ("seconds", args.seconds), ("freq_res", args.freq_res), ("dm", args.dm), ("coherent", args.coherent),
                         ("freq", args.freq)):
            if arg is not None:
                spec[key] = arg
    return spec


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


# kernels whose streaming reads FETCH_SIZE tallies at half their bytes (gfx950: 128-byte requests counted as 64, MI355X_MICROARCH.md
# "HBM").  frbch_k2_priv reads 8 bytes per lane, 64-byte runs of lines two waves share: calibrated against its known byte count
# (tools/pmc.sh, round 4: FETCH_SIZE = 0.50 - 0.56 of the spill it reads), so it takes the same factor
WIDE_FETCH = ("frbch_k2_", "frbch_k2c_", "frbch_quantise", "frbch_stats_partial", "frbch_k0_stage", "frbch_k1_wave", "frbch_k1_split")


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
    def why(msg):     # roofline.traffic is null then; say why on stderr
        print("bench.py: live PMC pass: " + msg, file=sys.stderr)
        return None
    if not shutil.which("rocprofv3"):
        return why("rocprofv3 not found")
    child = [a for a in argv if a not in ("--pmc-child",)]
    res = {}
    tmp = tempfile.mkdtemp(prefix="frbch_pmc_", dir="/tmp")
    try:
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, cnt)
            cmd = ["rocprofv3", "--pmc", cnt, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__)] + child + ["--pmc-child"]
            env = dict(os.environ, TMPDIR="/tmp")
            # own process group: on a timeout the profiler AND the python it started (which holds the GPU) are killed
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                # (the first `import torch` on a fresh box alone can take 1 - 2 minutes; a pass takes 15 - 30 s after that.  On one box
                # of the pool the FETCH_SIZE pass hung behind the HSA initialisation -- the limit is what such a box costs the bench)
                rc = pr.wait(timeout=300 if cnt == "FETCH_SIZE" else 180)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return why(cnt + ": timed out")
            # (a profiler that fails at EXIT has still written its counters: judge by the files, not by the status)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return why("%s: no counter file (profiler exit status %s)" % (cnt, rc))
            for path in files:
                for row in csv.DictReader(open(path)):
                    if row.get("Counter_Name") != cnt or "frbch" not in row.get("Kernel_Name", ""):
                        continue
                    k = row["Kernel_Name"]
                    rec = res.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {}})
                    rec[cnt] += float(row["Counter_Value"])
                    rec["n"][cnt] = rec["n"].get(cnt, 0) + 1
    except Exception as exc:
        return why(repr(exc))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"_source": "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (KB counters; wide fetches x2)"}
    total = 0.0
    for k, rec in res.items():
        short = short_kernel_name(k)
        if not short:
            continue
        nl = max(1, min(rec["n"].get("FETCH_SIZE", 1), rec["n"].get("WRITE_SIZE", 1)))
        corr = 2.0 if short.startswith(WIDE_FETCH) else 1.0
        out[short] = (rec["FETCH_SIZE"] / max(1, rec["n"].get("FETCH_SIZE", 1)) * corr +
                      rec["WRITE_SIZE"] / max(1, rec["n"].get("WRITE_SIZE", 1))) * 1024.0
        out.setdefault("_launches", {})[short] = nl
        total += (rec["FETCH_SIZE"] * corr + rec["WRITE_SIZE"]) * 1024.0
    # every frbch kernel of the child's two steps (one warm-up + one timed: --steps 1 --warmup 1), whatever its name: per step
    out["_step_total"] = total / 2.0
    return out


PCIE_PEAK_GBS = 64.0   # PCIe Gen5 x16, one direction (MI355X_MICROARCH.md host link)


def budget_bytes_per_sample(spec):
    """0.502 in + 8 + 8 spill + output bytes per dual-pol sample (8-bit codes: nprod / (2 T)); the coherent
    path spills twice"""
    nprod = 4 if spec["pol"] >= 4 else 1
    return 0.502 + 16.0 * (2 if spec["coherent"] else 1) + nprod / (2.0 * spec["tscrunch"])


class Workload:
    """`nif` IFs of one BASELINE configuration on one GPU: handles, synthetic frames and the row buffer, all resident in HBM"""

    def __init__(self, torch, dev, spec, args, local_rank, rank):
        from frb_baseband_amd import channeliser as ch
        self.torch, self.spec, self.ch = torch, spec, ch
        nif = spec["nif"]
        self.cfg_kwargs = []
        self.chans = []
        world = int(os.environ.get("WORLD_SIZE", "1") or 1)
        for i in range(nif):
            # IF numbering and sidebands as base2fil.sh:30-67 (odd IFs LSB, even USB); rows in splice order: highest IF first
            ifno = nif - i
            # sky frequencies: the IFs of the node tile one band around spec["freq"], descending in splice order, rank 0 on top
            # (base2fil.sh:54,65,254) -- frbch_join checks that the ranks' pieces continue each other in frequency
            g = rank * nif + i
            freq = spec["freq"] if spec["coherent"] else spec["freq"] + ((world * nif - 1) / 2.0 - g) * spec["bw"]
            kw = dict(bw_mhz=-spec["bw"] if ifno % 2 else spec["bw"], nchan=spec["nchan"], pol_mode=spec["pol"], nbit_out=8,
                      tscrunch=spec["tscrunch"], rescale_constant=1, rescale_interval_s=10.0, total_s=spec["seconds"] * spec.get("tile", 1),
                      device=local_rank, max_blocks_per_launch=args.maxb, flags=args.flags, dm=spec["dm"],
                      coherent=1 if spec["coherent"] else 0, freq_mhz=freq, freq_res=spec["freq_res"], overlap=args.overlap)
            if nif == 1:
                kw["bw_mhz"] = spec["bw"]
            self.cfg_kwargs.append(kw)
            self.chans.append(ch.Channeliser(ch.new_config(**kw)))
        self.info = self.chans[0].info
        self.frames = []
        for i in range(nif):
            fr, self.nfr = synth_frames_device(torch, dev, spec["seconds"], spec["bw"], spec["nchan"], if_index=rank * nif + (nif - i))
            if spec.get("tile", 1) > 1:   # a long scan as repetitions of the synthesised seconds (the device path takes frames as they are)
                fr = fr.repeat(spec["tile"])
                self.nfr *= spec["tile"]
            self.frames.append(fr)
        torch.cuda.synchronize()   # the library runs on its own streams: inputs must be complete
        info = self.info
        self.nblocks = (self.nfr * 8000 - info.block_payload_bytes) // info.block_stride_bytes + 1   # overlap-save when coherent
        self.rows = self.nblocks * info.rows_per_block
        self.row_pitch = nif * info.row_bytes
        self.out = torch.empty(self.rows * self.row_pitch, dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.samples_per_step = nif * self.nblocks * info.block_stride_bytes * 2   # new dual-pol samples consumed (2-bit: 2 per byte)
        self.ptrs = [f.data_ptr() for f in self.frames]

    def scan(self, flush=True):
        from frb_baseband_amd import multi_if
        return multi_if.scan_device(self.chans, self.ptrs, self.nfr, 8032, 32, 0, self.nblocks, self.out.data_ptr(), self.rows,
                                    flush=flush, stream=self.stream)

    def step(self):
        for c in self.chans:
            c.reset()     # (waits for the previous step's work before it touches the rescale state)
        return self.scan()

    def timing(self):
        """per-kernel device time, launches and algorithmic bytes summed over the IFs"""
        agg = {}
        for c in self.chans:
            for k, v in c.get_timing().items():
                a = agg.setdefault(k, {"launches": 0, "total_ms": 0.0, "algorithmic_bytes": 0.0})
                for f in a:
                    a[f] += v[f]
        return agg

    def measure(self, steps, warmup, dist=None):
        torch = self.torch
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        for c in self.chans:
            c.set_profiling(True)
            c.timing_reset()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            got = self.step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        assert got == self.rows, (got, self.rows)
        return dt

    def steady_state(self, steps):
        """extra (not `value`): steady state of a long scan -- scale frozen after the first interval (-c), K2 digitises in-kernel"""
        torch = self.torch
        for c in self.chans:
            c.set_profiling(False)
            off, sc = c.get_rescale()
            c.reset()
            c.set_rescale(off, sc)
        self.scan(flush=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            self.scan(flush=False)
        torch.cuda.synchronize()
        return self.samples_per_step * steps / (time.perf_counter() - t1) / 1e6

    def release_device(self):
        """handles, frames and row buffer freed (geometry and configuration stay readable)"""
        for c in self.chans:
            c.close()
        self.chans = []
        self.out = None
        self.frames = []
        self.torch.cuda.empty_cache()

    def close(self):
        self.release_device()

    def describe(self, world):
        sp, info = self.spec, self.info
        prod = {2: "Stokes-I", 4: "coherency (-d4)", 5: "full-Stokes IQUV (-d4 products -> I,Q,U,V)"}.get(sp["pol"], "pol%d" % sp["pol"])
        dflag = {2: "-d1", 3: "-d3", 4: "-d4", 5: "-d4 -iquv"}.get(sp["pol"], "-P%d" % sp["pol"])
        return (f"{sp['nif']} IF x {sp['bw']:g} MHz 2-bit dual-pol VDIF -> {sp['nchan']}-ch {prod} 8-bit .fil rows "
                f"(per IF: -c -b8 {dflag}{' -t %d' % sp['tscrunch'] if sp['tscrunch'] > 1 else ''} -F{sp['nchan']}:{info.freq_res}"
                f"{' -D %g -F%d:D' % (sp['dm'], sp['nchan']) if sp['coherent'] else ''}), {sp['seconds'] * sp.get('tile', 1):g} s per IF per step, "
                f"{sp['nif']} IF per GPU on {world} GPU(s) (no data-path collective), one step = all IFs of the GPU into one "
                f"[row][product][IF-major channel] row buffer (frbch_scan_device), first rescale interval measured every step, "
                f"frames and rows resident in HBM")


def budget_share_bytes_per_sample(name, spec):
    """a kernel's share of the SURVEY 8(d) budget, bytes per dual-pol sample: the compulsory input and the spill write belong to
    K1, the spill read and the output codes to K2 (the coherent path spills twice: K2c reads and writes, K3 reads, K4 emits);
    K0's corner-turned copy, the statistics and the digitiser of a buffered rescale interval move bytes the budget does not hold"""
    nprod = 4 if spec["pol"] >= 4 else 1
    out = nprod / (2.0 * spec["tscrunch"])
    if name.startswith("frbch_k1_"):
        return 0.502 + 8.0
    if name.startswith("frbch_k2c_"):
        return 16.0
    if name.startswith("frbch_k2_"):
        return 8.0 + (0.0 if spec["coherent"] else out)
    if name.startswith("frbch_k3_"):
        return 8.0
    if name.startswith("frbch_k4_"):
        return out
    return 0.0


def roofline_of(timing, spec, samples):
    """the dominant kernel = the largest summed launch time, no exclusions; `samples` = dual-pol samples of the measured steps.
    Returns (name, record, GB/s of its 8(d) share, GB/s of the engine's own per-launch byte model)"""
    name, rec = max(timing.items(), key=lambda kv: kv[1]["total_ms"])
    sec = rec["total_ms"] * 1e-3
    share = budget_share_bytes_per_sample(name, spec) * samples / sec / 1e9 if sec > 0 else 0.0
    model = rec["algorithmic_bytes"] / sec / 1e9 if sec > 0 else 0.0
    return name, rec, share, model


def kernels_overlap(timing, steps, dt):
    """True when the per-kernel HIP-event times add up to visibly more than the step: kernels shared the chip"""
    return sum(v["total_ms"] for v in timing.values()) > 1.02 * dt * 1e3


def concurrency_note(timing, steps, dt):
    if not kernels_overlap(timing, steps, dt):
        return "kernels run one after the other on the whole chip"
    return ("kernels of consecutive IFs of a scan share the chip on plain streams (DESIGN.md section 4b): a kernel's launch duration is "
            "its time on ITS share of the chip and of HBM, and the sum over kernels exceeds ms_per_step")


def step_traffic(live_traffic, timing, steps):
    """HBM bytes of all kernels of ONE step: live PMC bytes per launch (collect_traffic) x launches per step of the timed run"""
    if not live_traffic:
        return None, None
    total, missing = 0.0, []
    for k, v in timing.items():
        if not v["launches"]:
            continue
        if k in live_traffic:
            total += live_traffic[k] * v["launches"] / steps
        else:
            missing.append(k)
    if "_step_total" in live_traffic:      # the sum over every kernel of the profiled steps (timing-slot names need not match the profiler's)
        return live_traffic["_step_total"], []
    return total, missing


def host_inclusive(args, torch, dist, wl, rank, world):
    """What the reference's calls do end to end: VDIF files -> .fil, PCIe both ways and the host threads included; files on
    tmpfs (disk excluded).  One IF through frbch_run_file (process_vdif.py:191 `digifil ... -o out hdr`), and -- when the GPU
    holds several IFs -- the whole scan through frbch_run_scan (N digifil + N FIFOs + splice of base2fil.sh:348-350,404-448 in
    one call) into /dev/null (a sink that does not copy: the library's own pipeline) ."""
    ch = wl.ch
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    nif = wl.spec["nif"] if world == 1 else min(wl.spec["nif"], 2)    # (N ranks share one host: two IFs per rank in this leg)
    vds = [os.path.join(base, f"frbch_bench_{os.getpid()}_r{rank}_if{i}.vdif") for i in range(nif)]
    fil = vds[0].replace(".vdif", ".fil")
    res = {}
    try:
        for i in range(nif):
            wl.frames[i][: wl.nfr * 8032].cpu().numpy().tofile(vds[i])
        in_bytes = os.path.getsize(vds[0])
        rows_total, row_bytes_if, dev_out = wl.rows, wl.info.row_bytes, wl.out.device
        wl.release_device()      # the timed workload's handles and buffers go: the legs below open their own (8 x ~15 GB with four products)
        per_if = wl.samples_per_step // wl.spec["nif"]

        def timed(fn):
            times = []
            for i in range(args.host_runs + 1):          # first pass untimed: pinned buffers, page cache, clocks
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                fn()
                dt = time.perf_counter() - t0
                if dist is not None:
                    t = torch.tensor([dt], dtype=torch.float64, device=dev_out if args.backend == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                if i:
                    times.append(dt)
            return sorted(times)[len(times) // 2]

        with ch.Channeliser(ch.new_config(**wl.cfg_kwargs[0])) as c:
            def one():
                c.reset()
                if os.path.exists(fil):
                    os.remove(fil)        # what run_digifil does with --force before the call (process_vdif.py:146-149)
                c.run_file(vds[0], fil)
            best = timed(one)
        out_bytes = os.path.getsize(fil)
        each_way = max(in_bytes, out_bytes) / best / 1e9
        res["run_file"] = {"value": round(per_if * world / best / 1e6, 1), "unit": "Msamples/s", "s_per_scan": round(best, 4),
                           "realtime_x": round(per_if / best / 1e6 / (2 * wl.spec["bw"]), 1),
                           "bytes_in_per_if": in_bytes, "bytes_out_per_if": out_bytes,
                           "pcie": {"bound": "pcie", "achieved": round(each_way, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s per direction per GPU",
                                    "frac": round(each_way / PCIE_PEAK_GBS, 4)},
                           "path": f"frbch_run_file, one IF: {base} VDIF -> pinned ring -> HBM -> pinned ring -> {base} .fil, median of {args.host_runs} scans after one untimed"}
        if nif > 1:
            from frb_baseband_amd import multi_if
            chans = [ch.Channeliser(ch.new_config(**kw)) for kw in wl.cfg_kwargs[:nif]]
            try:
                def scan():
                    for c in chans:
                        c.reset()
                    multi_if.run_scan(chans, vds, "/dev/null")
                best = timed(scan)
            finally:
                for c in chans:
                    c.close()
            out_total = rows_total * row_bytes_if * nif
            each_way = max(in_bytes * nif, out_total) / best / 1e9
            res["run_scan"] = {"value": round(per_if * nif * world / best / 1e6, 1), "unit": "Msamples/s", "s_per_scan": round(best, 4),
                               "bytes_in": in_bytes * nif, "bytes_out": out_total,
                               "pcie": {"bound": "pcie", "achieved": round(each_way, 2), "peak": PCIE_PEAK_GBS,
                                        "unit": "GB/s in the busier direction per GPU", "frac": round(each_way / PCIE_PEAK_GBS, 4)},
                               "path": f"frbch_run_scan, {nif} IFs: {base} VDIF files -> pinned rings -> HBM (rows joined in the K2 / digitiser store addresses) -> pinned ring -> /dev/null"}
        if world == 1:
            # A scan of minutes, as the reference records them (frb.conf `lengths`): 60 s of ONE IF as Stokes I (-d1); the first
            # 10-s rescale interval is buffered, after it input, transform and output stream batch by batch.  Into /dev/null
            # (the library's own pipeline) and into a named FIFO drained by a reader (base2fil.sh:348-350: what digifil writes
            # into and splice reads from).
            import threading
            spec60 = dict(wl.spec, nif=1, pol=2, seconds=60.0)
            fr60, nfr60 = synth_frames_device(torch, dev_out, 60.0, spec60["bw"], spec60["nchan"], if_index=99)
            vd60 = os.path.join(base, f"frbch_bench_{os.getpid()}_60s.vdif")
            fifo = os.path.join(base, f"frbch_bench_{os.getpid()}_60s.fifo")
            try:
                fr60[: nfr60 * 8032].cpu().numpy().tofile(vd60)
                del fr60
                kw60 = dict(wl.cfg_kwargs[0], pol_mode=2, total_s=60.0, bw_mhz=spec60["bw"])
                samples60 = nfr60 * 8000 * 2
                os.mkfifo(fifo)
                with ch.Channeliser(ch.new_config(**kw60)) as c:
                    def to_null():
                        c.reset()
                        c.run_file(vd60, "/dev/null")

                    def to_fifo():
                        c.reset()

                        def drain():
                            with open(fifo, "rb", buffering=0) as f:
                                buf = bytearray(1 << 24)
                                while f.readinto(buf):
                                    pass
                        th = threading.Thread(target=drain)
                        th.start()
                        c.run_file(vd60, fifo)
                        th.join()
                    t_null = timed(to_null)
                    t_fifo = timed(to_fifo)
                    by_ref = bool(c.get_info().diag & 2)      # the rows went into the pipe by reference (vmsplice of the pinned ring)
                    os.environ["FRBCH_FIFO_COPY"] = "1"       # the same with plain write() calls
                    try:
                        t_fifo_copy = timed(to_fifo)
                    finally:
                        del os.environ["FRBCH_FIFO_COPY"]
                out60 = samples60 // 2       # 8-bit Stokes I: one byte per two dual-pol samples
                res["long_scan"] = {
                    "scan": f"60 s of one {spec60['bw']:g} MHz IF -> {spec60['nchan']}-ch Stokes I 8 bit (-c -I 10: first interval buffered, then streamed)",
                    "bytes_in": os.path.getsize(vd60), "bytes_out": out60,
                    "dev_null": {"value": round(samples60 / t_null / 1e6, 1), "unit": "Msamples/s", "s_per_scan": round(t_null, 4),
                                 "pcie_frac_each_way": round(out60 / t_null / 1e9 / PCIE_PEAK_GBS, 4)},
                    "fifo_drained": {"value": round(samples60 / t_fifo / 1e6, 1), "unit": "Msamples/s", "s_per_scan": round(t_fifo, 4),
                                     "pcie_frac_each_way": round(out60 / t_fifo / 1e9 / PCIE_PEAK_GBS, 4), "by_reference": by_ref,
                                     "sink_GBs": round(out60 / t_fifo / 1e9, 2)},
                    "fifo_drained_write_only": {"value": round(samples60 / t_fifo_copy / 1e6, 1), "unit": "Msamples/s",
                                                "s_per_scan": round(t_fifo_copy, 4), "sink_GBs": round(out60 / t_fifo_copy / 1e9, 2)}}
            finally:
                for f in (vd60, fifo):
                    try:
                        os.remove(f)
                    except OSError:
                        pass
        if world > 1 and dist is not None:
            # the node-level scan (python -m frb_baseband_amd.scan, base2fil.sh:30-67 + 348-350 + 404-448): every rank runs its
            # IFs through frbch_run_scan into a named FIFO, ONE native join (csrc/frbch_join, started by rank 0; it never touches
            # a GPU) concatenates the ranks' pieces into the IFall stream while they produce
            import subprocess
            from frb_baseband_amd import multi_if, scan as scan_mod
            tag = os.environ.get("MASTER_PORT", "0")
            fifos = [os.path.join(base, f"frbch_bench_join_{tag}_rank{r}.fil") for r in range(world)]
            # (the `nif` IFs of this leg tile one band across the ranks, rank 0 on top: frbch_join checks the pieces' continuity)
            bw_leg = abs(wl.cfg_kwargs[0]["bw_mhz"])
            chans = [ch.Channeliser(ch.new_config(**(kw if wl.spec["coherent"] else
                                                    dict(kw, freq_mhz=wl.spec["freq"] + ((world * nif - 1) / 2.0 - (rank * nif + i)) * bw_leg))))
                     for i, kw in enumerate(wl.cfg_kwargs[:nif])]
            try:
                def node_scan():
                    join = None
                    if rank == 0:
                        for f in fifos:
                            if os.path.exists(f):
                                os.remove(f)
                            os.mkfifo(f)
                        join = subprocess.Popen([scan_mod.JOIN_PATH, "/dev/null"] + fifos, stdout=subprocess.DEVNULL)
                    dist.barrier()
                    for c in chans:
                        c.reset()
                    multi_if.run_scan(chans, vds, fifos[rank])
                    if join is not None and join.wait(timeout=600) != 0:
                        raise RuntimeError("frbch_join failed")
                    dist.barrier()
                best = timed(node_scan)
            finally:
                for c in chans:
                    c.close()
                if rank == 0:
                    for f in fifos:
                        if os.path.exists(f):
                            os.remove(f)
            res["node_scan"] = {"value": round(per_if * nif * world / best / 1e6, 1), "unit": "Msamples/s", "s_per_scan": round(best, 4),
                                "n_gpus": world, "ifs": nif * world,
                                "path": f"{world} ranks x frbch_run_scan ({nif} IFs each) -> {world} FIFOs -> frbch_join -> /dev/null (one IFall stream of {nif * world * wl.spec['nchan']} channels)"}
        return res
    except Exception as exc:   # a reported extra: never fail the bench for it
        res["error"] = repr(exc)
        return res
    finally:
        for f in vds + [fil]:
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
    spec = workload_spec(args)

    # live PMC passes first: the children must start before this process initialises the GPU
    live_traffic = None
    if world == 1 and not args.no_traffic and not args.pmc_child:
        live_traffic = collect_traffic(argv + ["--steps", "1", "--warmup", "1", "--no-cpu", "--no-host", "--no-traffic", "--no-configs", "--no-steady"])
    if args.pmc_child:
        args.steps, args.warmup, args.no_cpu, args.no_host, args.no_configs = 1, 1, True, True, True

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

    wl = Workload(torch, dev, spec, args, local_rank, rank)
    dt = wl.measure(args.steps, args.warmup, dist)
    per_rank = None
    if dist is not None:
        red_dev = dev if args.backend == "nccl" else "cpu"
        tall = [torch.zeros(1, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], device=red_dev, dtype=torch.float64))
        per_rank = [float(t.item()) for t in tall]
        dt = max(per_rank)
    timing = wl.timing()
    if args.pmc_child:
        wl.close()
        return
    steady = wl.steady_state(args.steps) if not args.no_steady else 0.0
    samples_per_step, nblocks, nif = wl.samples_per_step, wl.nblocks, spec["nif"]
    workload_text = wl.describe(world)
    host = None
    if not args.no_host:
        host = host_inclusive(args, torch, dist, wl, rank, world)
    wl.close()

    # the other BASELINE configurations, short runs on this GPU (rank 0, N = 1): driver-visible, not `value`
    configs = None
    if world == 1 and not args.no_configs:
        configs = {}
        for name in ("cfg2", "cfg4", "cfg5", "cfg1", "online32"):
            if name == args.workload:
                continue
            try:
                sp = workload_spec(args, name)
                w2 = Workload(torch, dev, sp, args, local_rank, rank)
                d2 = w2.measure(args.config_steps, 3)
                t2 = w2.timing()
                st2 = w2.steady_state(args.config_steps)
                v2 = w2.samples_per_step * args.config_steps / d2 / 1e6
                nm, rec, ach, _model = roofline_of(t2, sp, w2.samples_per_step * args.config_steps)
                bps = budget_bytes_per_sample(sp)
                configs[name] = {"workload": w2.describe(1), "value": round(v2, 1), "unit": "Msamples/s", "steps": args.config_steps,
                                 "ms_per_step": round(d2 / args.config_steps * 1e3, 4), "steady_state": round(st2, 1),
                                 "frac": round(v2 * 1e6 * bps / 1e9 / HBM_PEAK_GBS, 4),      # whole path, SURVEY 8(d) budget
                                 "steady_state_frac": round(st2 * 1e6 * bps / 1e9 / HBM_PEAK_GBS, 4),
                                 "dominant": nm, "dominant_frac": round(ach / HBM_PEAK_GBS, 4),   # its share of that budget / its time
                                 "kernels_ms_per_step": {k: round(v["total_ms"] / args.config_steps, 4) for k, v in t2.items() if v["launches"]}}
                w2.close()
            except Exception as exc:   # reported extras: never fail the bench for them
                configs[name] = {"error": repr(exc)}
        # the timed workload as a scan of a MINUTE per IF (the reference's scans last minutes, frb.conf:11-16): the first 10-s
        # rescale interval buffered and digitised, the other 50 s with K2 digitising in-kernel -- what a real scan sees
        try:
            sp = dict(workload_spec(args), tile=6)      # 6 x the 10 synthesised seconds
            w2 = Workload(torch, dev, sp, args, local_rank, rank)
            d2 = w2.measure(2, 1)
            t2 = w2.timing()
            v2 = w2.samples_per_step * 2 / d2 / 1e6
            bps = budget_bytes_per_sample(sp)
            configs[args.workload + "_60s"] = {"workload": w2.describe(1), "value": round(v2, 1), "unit": "Msamples/s", "steps": 2,
                                               "ms_per_step": round(d2 / 2 * 1e3, 4),
                                               "frac": round(v2 * 1e6 * bps / 1e9 / HBM_PEAK_GBS, 4),
                                               "kernels_ms_per_step": {k: round(v["total_ms"] / 2, 4) for k, v in t2.items() if v["launches"]}}
            w2.close()
        except Exception as exc:
            configs[args.workload + "_60s"] = {"error": repr(exc)}

    if rank == 0:
        total_samples = samples_per_step * args.steps * world
        value = total_samples / dt / 1e6
        samples_timed = samples_per_step * args.steps                      # this rank's samples of the timed region
        name, rec, ach, model = roofline_of(timing, spec, samples_timed)
        bps = budget_bytes_per_sample(spec)
        whole = value / world * 1e6 * bps / 1e9                            # GB/s of SURVEY 8(d) bytes, per GPU
        # measured HBM bytes of ALL kernels of one step: live PMC passes of this run (collect_traffic), never a stored constant
        traffic, traffic_missing = step_traffic(live_traffic, timing, args.steps)
        valu = None
        if name in VALU_PER_WAVE_BLOCK and rec["total_ms"] > 0:     # the other roof of this kernel: fp32 VALU lane operations
            per_wave, waves = VALU_PER_WAVE_BLOCK[name]
            lane_ops = per_wave * waves * 64.0 * (2 * spec["nchan"] // 8) * nblocks * nif * args.steps   # (every block of every IF and step once)
            tl = lane_ops / (rec["total_ms"] * 1e-3) / 1e12
            valu = {"bound": "valu", "achieved": round(tl, 2), "peak": VALU_PEAK_TLOPS, "unit": "T lane-ops/s (fp32, unpacked)",
                    "frac": round(tl / VALU_PEAK_TLOPS, 4), "instructions_per_wave_and_block": per_wave}
        share = budget_share_bytes_per_sample(name, spec)
        roof = {"bound": "hbm", "scope": "whole path: value x SURVEY 8(d) budget bytes per sample (kernel_* = the dominant kernel with its share of it)",
                "achieved": round(whole, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(whole / HBM_PEAK_GBS, 5),
                "algorithmic_bytes_per_sample": round(bps, 3),
                "algorithmic_bytes_per_step": bps * samples_per_step,
                "traffic": traffic, "traffic_ratio": (round(traffic / (bps * samples_per_step), 4) if traffic else None),
                "traffic_source": (live_traffic or {}).get("_source"), "traffic_kernels_without_counters": traffic_missing,
                "kernel": name, "kernel_frac": round(ach / HBM_PEAK_GBS, 5), "kernel_achieved": round(ach, 2),
                "kernel_budget_bytes_per_sample": share,
                "kernel_avg_launch_ms": round(rec["total_ms"] / max(1, rec["launches"]), 5),
                "kernel_launches_per_step": rec["launches"] / args.steps,
                "kernel_budget_bytes_per_launch": share * samples_timed / max(1, rec["launches"]),
                "kernel_model_frac": round(model / HBM_PEAK_GBS, 5),     # the engine's own byte model of that kernel (float rows of a buffered interval included)
                "kernel_traffic_per_launch": (live_traffic or {}).get(name),
                "steady_state_frac": round(steady * 1e6 * bps / 1e9 / HBM_PEAK_GBS, 5),
                "valu_frac": (valu or {}).get("frac"),
                "dominance_rule": "largest summed launch time (HIP events on the launch streams), no exclusions",
                "concurrency": concurrency_note(timing, args.steps, dt),
                "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in timing.items() if v["launches"]},
                "per_kernel": {k: {"ms_per_step": round(v["total_ms"] / args.steps, 4),
                                   "budget_frac": round(budget_share_bytes_per_sample(k, spec) * samples_timed / (v["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "model_frac": round(v["algorithmic_bytes"] / (v["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "traffic_per_launch": (live_traffic or {}).get(k)}
                               for k, v in timing.items() if v["launches"] and v["total_ms"] > 0},
                "valu": valu}
        line = {
            "metric": "Msamples/s channelised to .fil per GPU; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text, "baseline_config": args.workload,
                       "samples_per_step_per_gpu": samples_per_step, "blocks_per_step_per_if": nblocks, "ifs_per_gpu": nif,
                       "realtime_x": round(value / world / (2 * spec["bw"] * nif), 2),
                       "steady_state_msamples_per_gpu": round(steady, 1),
                       "per_rank_seconds": per_rank},
            "roofline": roof,
            "host_inclusive": host,
            "configs": configs,
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, spec["bw"], spec["nchan"], spec["pol"])
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
