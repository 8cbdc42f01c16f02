cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3w
for a in "--workload cfg2" "--workload cfg2 --flags 2097152" "--workload cfg2" "--workload cfg2 --flags 2097152" "--workload cfg3 --steps 5" "--workload cfg3 --flags 2097152 --steps 5"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --warmup 3 > gpurun_out/r3w/b.json 2> gpurun_out/r3w/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3w/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
