"""One command for one scan on one node: what base2fil.sh does with `run_process_vdif` (one process_vdif / digifil per IF,
base2fil.sh:30-67), one named FIFO per IF (:348-350) and sigproc `splice` (:404-448), re-cut for a node of GPUs.

    python -m frb_baseband_amd.scan --experiment pr001a --st ef --scanname 001 --workdir /scratch0/u/pr001a \\
        --source R3 --ra 01:58:00.75 --dec 65:43:00.3 --station effelsberg --nif 16 --bw 64 --freqLSB_0 1340.49 \\
        --nchan 4096 --tscrunch 8 --nsec 10 --pol 2 --nbit 8 --outdir /data1/u/pr001a [--gpus 8]

The parameter names are frb.conf's (experiment, nif, bw, freqLSB_0, station, nchan, tscrunch, pol) and run_process_vdif's
(scanname, st, workdir, nsec, start, nbit, keepBP); the per-IF input is <workdir>/<experiment>_<st>_no0<scanname>_IF<i>.vdif and
the product <outdir>/<experiment>_<st>_no0<scanname>_IFall_vdif_pol<pol>.fil (base2fil.sh:61,389), unless given explicitly.

This process never touches a GPU.  Before any GPU call exists it starts

  * one RANK process per GPU (`python -m frb_baseband_amd.scan --rank r ...`): it owns a contiguous run of the splice order
    (IF i -> GPU by position: 16 IFs on 8 GPUs = 2 IF per GPU, BASELINE configs[3]) and runs frbch_run_scan over its
    share -- its IFs are joined on its GPU, in the store addresses of the last kernel -- into a named FIFO;
  * one `frbch_join` (csrc/frbch_join.cpp): the streaming frequency concatenation of the ranks' FIFOs into the IFall file,
    strictly sequential, fed while the ranks produce.

There is no collective on the data path and no torch.distributed: ranks only meet in the join.  A failing rank (or join)
fails the run: the others are terminated and the exit status is non-zero.  With one GPU the rank writes the product itself.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import tempfile
import time

from . import multi_if

_HERE = os.path.dirname(os.path.abspath(__file__))
JOIN_PATH = os.path.join(_HERE, "csrc", "frbch_join")


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m frb_baseband_amd.scan", description=__doc__.split("\n\n")[0])
    ap.add_argument("--experiment", default="exp")
    ap.add_argument("--st", default="st", help="two-letter station code of the file names")
    ap.add_argument("--scanname", default="001")
    ap.add_argument("--workdir", default=".", help="where the per-IF VDIF files are")
    ap.add_argument("--vdif", nargs="*", default=None, help="per-IF VDIF files, IF1 IF2 ... (instead of the naming convention)")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--out", default=None, help="product path (default: <outdir>/<exp>_<st>_no0<scan>_IFall_vdif_pol<pol>.fil)")
    ap.add_argument("--source", default="unknown")
    ap.add_argument("--ra", default="00:00:00.0")
    ap.add_argument("--dec", default="00:00:00.0")
    ap.add_argument("--station", default="ONSALA85", help="tempo2 name of the telescope (process_vdif -t)")
    ap.add_argument("--nif", type=int, required=True)
    ap.add_argument("--bw", type=float, required=True, help="bandwidth per IF, MHz")
    ap.add_argument("--freqLSB_0", type=float, required=True, help="centre frequency of the lowest LSB IF, MHz (frb.conf)")
    ap.add_argument("--nchan", type=int, required=True)
    ap.add_argument("--tscrunch", type=int, default=1)
    ap.add_argument("--nsec", type=float, default=120.0)
    ap.add_argument("--start", type=float, default=0.0)
    ap.add_argument("--pol", type=int, default=2, help="as process_vdif --pol; 5 = Stokes I,Q,U,V from the -d4 products")
    ap.add_argument("--nbit", type=int, default=8)
    ap.add_argument("--keepBP", action="store_true")
    ap.add_argument("--gpus", type=int, default=0, help="GPUs of this node to use (0 = all visible)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--rank", type=int, default=None, help=argparse.SUPPRESS)      # set by the launcher
    ap.add_argument("--world", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--piece", default=None, help=argparse.SUPPRESS)
    return ap


def vdif_paths(args):
    if args.vdif:
        if len(args.vdif) != args.nif:
            raise SystemExit(f"--vdif lists {len(args.vdif)} files for --nif {args.nif}")
        return {i + 1: p for i, p in enumerate(args.vdif)}
    return {i: os.path.join(args.workdir, f"{args.experiment}_{args.st}_no0{args.scanname}_IF{i}.vdif") for i in range(1, args.nif + 1)}


def product_path(args):
    return args.out or os.path.join(args.outdir, multi_if.ifall_name(args.experiment, args.st, args.scanname, args.pol))


def rank_share(nif: int, world: int, rank: int):
    """contiguous runs of the splice order (highest IF first, base2fil.sh:350,367), so that the ranks' pieces concatenate"""
    order = multi_if.splice_order(nif)
    per = (nif + world - 1) // world
    return order[rank * per:(rank + 1) * per]


def run_rank(args) -> int:
    """one GPU's share of the scan: its IFs through frbch_run_scan into `args.piece` (a FIFO the join reads, or the product)"""
    from . import channeliser as ch
    from . import digifil_args
    from . import process_vdif as pv
    mine = rank_share(args.nif, args.world, args.rank)
    if not mine:
        return 0
    plans = {p.index: p for p in multi_if.plan_ifs(args.nif, args.freqLSB_0, args.bw)}
    paths = vdif_paths(args)
    device = 0 if args.share_gpu else args.rank
    chans, files = [], []
    try:
        for i in mine:
            p = plans[i]
            hdr = pv.make_hdr(args.source, p.freq_mhz, paths[i], pol=args.pol, usb=(p.sideband == "u"), ra=args.ra, dec=args.dec,
                              bw=args.bw, telescope=args.station)
            cmd = pv.digifil_command(hdr, args.piece, args.start, args.nsec, args.nchan, min(args.pol, 4), args.nbit, args.tscrunch,
                                     1, 0.0, False, args.keepBP, iquv=(args.pol == 5))
            cfg, _h, _o = digifil_args.parse(cmd)
            cfg.device = device
            chans.append(ch.Channeliser(cfg))
            files.append(paths[i])
        multi_if.run_scan(chans, files, args.piece)
    except ch.Error as exc:
        print(f"frb_baseband_amd.scan: rank {args.rank} (IFs {mine}): {exc}", file=sys.stderr)
        return 1
    finally:
        for c in chans:
            c.close()
    return 0


def count_gpus() -> int:
    """visible GPUs without initialising one (this process never touches the GPU)"""
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:
        return len([x for x in vis.split(",") if x.strip() != ""])
    try:
        n = len([d for d in os.listdir("/sys/class/kfd/kfd/topology/nodes")
                 if open(f"/sys/class/kfd/kfd/topology/nodes/{d}/gpu_id").read().strip() not in ("", "0")])
        return max(n, 1)
    except OSError:
        return 1


def launch(args, argv) -> int:
    world = args.gpus or count_gpus()
    world = max(1, min(world, args.nif))
    out = product_path(args)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    if os.path.exists(out) and not os.path.isfile(out):
        pass                                   # a pre-made FIFO as the product: written sequentially, never unlinked
    base = [sys.executable, "-m", "frb_baseband_amd.scan"] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.perf_counter()
    if world == 1:
        rc = subprocess.call(base + ["--rank", "0", "--world", "1", "--piece", out], env=env)
        if rc == 0:
            print(f"frb_baseband_amd.scan: {out} ({time.perf_counter() - t0:.2f} s, 1 GPU)")
        return 1 if rc else 0
    if not os.path.exists(JOIN_PATH):
        print(f"frb_baseband_amd.scan: {JOIN_PATH} missing: build the native pieces first (make -C frb_baseband_amd/csrc)", file=sys.stderr)
        return 1
    tmp = tempfile.mkdtemp(prefix="frbch_scan_")
    fifos = [os.path.join(tmp, f"rank{r}.fil") for r in range(world) if rank_share(args.nif, world, r)]
    procs = []
    try:
        for f in fifos:
            os.mkfifo(f)
        join = subprocess.Popen([JOIN_PATH, out] + fifos, stdout=subprocess.DEVNULL)
        procs.append(("join", join))
        for r, f in enumerate(fifos):
            procs.append((f"rank {r}", subprocess.Popen(base + ["--rank", str(r), "--world", str(world), "--piece", f], env=env)))
        failed = None
        alive = {name: p for name, p in procs}
        while alive and failed is None:
            for name, p in list(alive.items()):
                rc = p.poll()
                if rc is None:
                    continue
                del alive[name]
                if rc != 0:
                    failed = (name, rc)
            time.sleep(0.02)
        if failed is not None:
            print(f"frb_baseband_amd.scan: {failed[0]} failed (exit status {failed[1]}): terminating the others", file=sys.stderr)
            for name, p in alive.items():
                p.terminate()
            # a rank that died before it opened its FIFO leaves the join blocked in open(): unblock it
            for f in fifos:
                try:
                    fd = os.open(f, os.O_WRONLY | os.O_NONBLOCK)
                    os.close(fd)
                except OSError:
                    pass
            for name, p in alive.items():
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return 1
        print(f"frb_baseband_amd.scan: {out} ({time.perf_counter() - t0:.2f} s, {len(fifos)} GPUs)")
        return 0
    finally:
        for f in fifos:
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.rank is not None:
        return run_rank(args)
    return launch(args, argv)


if __name__ == "__main__":
    sys.exit(main())
