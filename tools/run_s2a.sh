cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2a
tools/micro/cvt_probe > gpurun_out/s2a/cvt_probe.txt 2>&1; echo "probe rc=$?"
cat gpurun_out/s2a/cvt_probe.txt
for a in "--workload cfg3" "--workload cfg2"; do
timeout -k 10 300 python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 10 --warmup 3 > gpurun_out/s2a/b.json 2> gpurun_out/s2a/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/s2a/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
