cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2b
timeout -k 10 600 python3 -m pytest tests/test_gpu_pins.py -m gpu -x -q -k "bit_exact or scan_device or config3" > gpurun_out/s2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2b/pytest.log
for a in "--workload cfg3" "--workload cfg2"; do
timeout -k 10 300 python3 bench.py $a --no-cpu --no-traffic --no-configs --no-host --steps 10 --warmup 3 > gpurun_out/s2b/b.json 2> gpurun_out/s2b/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/s2b/b.json').read().strip().splitlines()[-1]); print('$a', d['value'], d['config']['steady_state_msamples_per_gpu'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
bash tools/overlap_sweep.sh s2b cfg3 "33554592 33554608 33554624 33554640 33554656"
