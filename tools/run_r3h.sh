mkdir -p gpurun_out/r3h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "node_scan" > gpurun_out/r3h/pytest_node.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3h/pytest_node.log
timeout -k 10 300 python3 bench.py --no-cpu --no-traffic --no-configs --steps 5 --warmup 2 > gpurun_out/r3h/bench_host.json 2> gpurun_out/r3h/bench_host.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r3h/bench_host.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['host_inclusive'])[:1800])"
timeout -k 10 400 python3 bench.py --gpus 2 --share-gpu --nif 2 --no-cpu --no-traffic --no-configs --steps 3 --warmup 1 > gpurun_out/r3h/bench_2rank.json 2> gpurun_out/r3h/bench_2rank.err; echo "bench2 rc=$?"; tail -3 gpurun_out/r3h/bench_2rank.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3h/bench_2rank.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], json.dumps(d['host_inclusive'])[:2500])"
