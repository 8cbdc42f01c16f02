#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for a in "" "--flags 2097152"; do
python3 bench.py --no-cpu --no-host --no-traffic --steps 10 --warmup 5 $a 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms_per_step']
print('$a', 'value', j['value'], 'steady', j['config'].get('steady_state_msamples_per_gpu'), {a:b for a,b in k.items() if b>0})"
done; done
