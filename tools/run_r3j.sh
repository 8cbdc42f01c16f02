mkdir -p gpurun_out/r3j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "32- or 64- or lane_kernel or 128-" > gpurun_out/r3j/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3j/pytest.log
for c in 32 64; do
for bw in 32 64; do
python3 bench.py --workload cfg1 --nchan $c --bw $bw --no-cpu --no-traffic --no-configs --no-host --steps 5 --warmup 2 > gpurun_out/r3j/c${c}_$bw.json 2> gpurun_out/r3j/c${c}_$bw.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3j/c${c}_$bw.json').read().strip().splitlines()[-1]); print($c, 'bw', $bw, d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done; done
