#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "4096" > gpurun_out/k1w5_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/k1w5_tests.log
for a in "--nchan 4096 --bw 64 --seconds 5" "--nchan 4096 --bw 64 --seconds 5 --flags 8" ""; do
python3 bench.py --no-cpu --no-host --no-traffic --steps 10 --warmup 5 $a 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms_per_step']
print('$a', 'value', j['value'], {a:b for a,b in k.items() if b>0})"
done
