cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3x
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "32- or 64- or lane_kernel" > gpurun_out/r3x/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r3x/pytest.log
for a in "--workload cfg1 --nchan 32 --bw 32" "--workload cfg1 --nchan 32 --bw 32 --flags 33554432" "--workload cfg1 --nchan 32 --bw 32 --pol 5"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 8 --warmup 3 > gpurun_out/r3x/b.json 2> gpurun_out/r3x/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3x/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
