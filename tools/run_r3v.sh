cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3v
for a in "--workload cfg4" "--workload cfg4 --maxb 19" "--workload cfg5" "--workload cfg5 --maxb 38" "--workload cfg2 --tscrunch 16" "--workload cfg2 --nchan 2048" "--workload cfg2 --nchan 512"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 8 --warmup 3 > gpurun_out/r3v/b.json 2> gpurun_out/r3v/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3v/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['config']['blocks_per_step_per_if'], d['roofline']['kernels_ms_per_step'])"
done
