mkdir -p gpurun_out/r3p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "coherent or dm" > gpurun_out/r3p/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3p/pytest.log
for a in "--workload cfg5" "--workload cfg5 --flags 33554432" "--workload cfg5 --maxb 38" "--workload cfg5 --maxb 19"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 5 --warmup 2 > gpurun_out/r3p/b.json 2> gpurun_out/r3p/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3p/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
