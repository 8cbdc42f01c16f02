cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2e
timeout -k 10 600 python3 -m pytest tests/test_gpu_pins.py -m gpu -x -q -k "bit_exact" > gpurun_out/s2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2e/pytest.log
show() { python3 -c "
import json,sys; d=json.loads(open('gpurun_out/s2e/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['config'].get('steady_state_msamples_per_gpu'), d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$1"; }
plain="--no-cpu --no-traffic --no-configs --no-host"
timeout -k 10 300 python3 bench.py --workload cfg3 $plain --steps 10 --warmup 3 > gpurun_out/s2e/b.json 2> gpurun_out/s2e/b.err; show "product cfg3"
export FRBCH_LIB=$GRAFT_REPO_ROOT/frb_baseband_amd/csrc/libfrbch_exp.so
for t in 0 16 272 528 1040 2064 4112; do
FRBCH_TILE_PAD=$t timeout -k 10 300 python3 bench.py --workload cfg2 $plain --steps 10 --warmup 3 > gpurun_out/s2e/b.json 2> gpurun_out/s2e/b.err; show "cfg2 tile_pad $t"
done
for r in 63 95 127 191 193 255 383 511; do
FRBCH_QUANT_RP=$r timeout -k 10 300 python3 bench.py --workload cfg3 $plain --no-steady --steps 6 --warmup 2 > gpurun_out/s2e/b.json 2> gpurun_out/s2e/b.err; show "cfg3 rp $r"
done
