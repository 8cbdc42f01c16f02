cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3y
timeout -k 10 900 python3 -m pytest tests/test_gpu_pins.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not full_length and not parseval" > gpurun_out/r3y/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3y/pytest.log
for a in "--workload cfg3" "--workload cfg2"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 10 --warmup 3 > gpurun_out/r3y/b.json 2> gpurun_out/r3y/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3y/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], sum(d['roofline']['kernels_ms_per_step'].values()))"
done
