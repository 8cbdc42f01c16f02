mkdir -p gpurun_out/r3n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --no-cpu --no-traffic --no-configs --steps 5 --warmup 2 > gpurun_out/r3n/bench_host.json 2> gpurun_out/r3n/bench_host.err; echo "bench rc=$?"; tail -3 gpurun_out/r3n/bench_host.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r3n/bench_host.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['host_inclusive'])[:3000])"
