cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2k
timeout -k 10 900 python3 -m pytest tests/test_gpu_pins.py tests/test_gpu_parity.py -m gpu -x -q -k "not power_error" > gpurun_out/s2k/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2k/pytest.log
plain="--no-cpu --no-traffic --no-configs --no-host"
for a in "--workload cfg3" "--workload cfg2"; do
timeout -k 10 300 python3 bench.py $a $plain --steps 10 --warmup 3 > gpurun_out/s2k/b.json 2> gpurun_out/s2k/b.err
python3 -c "
import json,sys; d=json.loads(open('gpurun_out/s2k/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['config'].get('steady_state_msamples_per_gpu'), d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" "$a"
done
