mkdir -p gpurun_out/r3o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "4096" > gpurun_out/r3o/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3o/pytest.log
for a in "--workload cfg4 --pol 5" "--workload cfg4 --pol 5 --flags 2" "--workload cfg3" "--workload cfg3 --maxb 76"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 5 --warmup 2 > gpurun_out/r3o/b.json 2> gpurun_out/r3o/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3o/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
