mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "32- or 64- or lane_kernel or 128-" > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3i/pytest.log
for c in 32 64 128; do
for f in 0 2; do
python3 bench.py --workload cfg1 --nchan $c --bw 32 --flags $f --no-cpu --no-traffic --no-configs --no-host --steps 5 --warmup 2 > gpurun_out/r3i/c${c}_$f.json 2> gpurun_out/r3i/c${c}_$f.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3i/c${c}_$f.json').read().strip().splitlines()[-1]); print($c, 'flags', $f, d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done; done
timeout -k 10 300 python3 bench.py --no-cpu --no-traffic --no-configs --steps 5 --warmup 2 > gpurun_out/r3i/bench_host.json 2> gpurun_out/r3i/bench_host.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r3i/bench_host.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['host_inclusive'])[:1800])"
