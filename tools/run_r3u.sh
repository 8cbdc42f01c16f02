cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "32- or 64- or lane_kernel or 128-" > gpurun_out/r3u/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3u/pytest.log
for a in "--workload cfg1 --nchan 32 --bw 32" "--workload cfg1 --nchan 64 --bw 32" "--workload cfg1"; do
python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 8 --warmup 3 > gpurun_out/r3u/b.json 2> gpurun_out/r3u/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3u/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
