#!/bin/bash
# usage: tools/ab.sh <other libfrbch.so> [bench args]: the same bench with the in-tree library and with another build, alternating (same box)
other=$1; shift
run() {
  python3 -c "
import sys
sys.argv = ['bench.py'] + sys.argv[1:]
from frb_baseband_amd import _lib
if '$1': _lib.LIB_PATH = '$1'
import bench
bench.main()" --no-cpu --no-host --no-traffic --steps 10 --warmup 5 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms_per_step']
print('$1' or 'in-tree', j['value'], {a:b for a,b in k.items() if b>0})"
}
for i in 1 2 3; do run "" "$@"; run "$other" "$@"; done
