#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py -m gpu -q -x > gpurun_out/t.log 2>&1; echo rc=$?; tail -8 gpurun_out/t.log
for a in "--nchan 2048 --freq-res 4096 --dm 56.7 --coherent --freq 1400" "--nchan 2048 --freq-res 4096 --dm 56.7 --coherent --freq 1400 --flags 8"; do
python3 bench.py --no-cpu --no-host --no-traffic --steps 10 --warmup 5 $a 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms_per_step']
print('$a', 'value', j['value'], 'steady', j['config'].get('steady_state_msamples_per_gpu'), {a:b for a,b in k.items() if b>0})"
done
