#!/bin/bash
# usage: [ARGS='--nchan 4096 --bw 64 --seconds 5'] tools/ablate.sh <tag> <flag values...>   (EXPERIMENTS build on the GPU box: bench per ablation mask)
tag=$1; shift
for f in "$@"; do
  python3 bench.py --no-cpu --no-host --no-traffic --steps 10 --warmup 5 $ARGS --flags $f 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms_per_step']
print('flags', $f, 'value', j['value'], {a:b for a,b in k.items() if b>0})"
done
