cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2m
plain="--no-cpu --no-traffic --no-configs --no-host"
for rep in 1 2; do
for lib in libfrbch.so libfrbch_alt.so; do
FRBCH_LIB=$GRAFT_REPO_ROOT/frb_baseband_amd/csrc/$lib timeout -k 10 300 python3 bench.py --workload cfg3 $plain --steps 10 --warmup 3 > gpurun_out/s2m/b.json 2> gpurun_out/s2m/b.err
python3 -c "
import json,sys; d=json.loads(open('gpurun_out/s2m/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['config'].get('steady_state_msamples_per_gpu'), d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$lib"
done
done
