cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2f
timeout -k 10 60 tools/micro/anyorder_probe > gpurun_out/s2f/anyorder.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/s2f/anyorder.txt
plain="--no-cpu --no-traffic --no-configs --no-host"
for a in cfg3 cfg2; do
timeout -k 10 300 python3 bench.py --workload $a $plain --steps 10 --warmup 3 > gpurun_out/s2f/b.json 2> gpurun_out/s2f/b.err
python3 -c "
import json,sys; d=json.loads(open('gpurun_out/s2f/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['config'].get('steady_state_msamples_per_gpu'), d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$a"
done
