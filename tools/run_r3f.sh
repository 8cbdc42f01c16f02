mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_pins.py tests/test_cornerturn.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r3f/pytest.log
timeout -k 10 300 python3 bench.py --no-cpu --no-traffic --no-configs --steps 5 --warmup 2 > gpurun_out/r3f/bench_host.json 2> gpurun_out/r3f/bench_host.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r3f/bench_host.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['host_inclusive'])[:1500])"
