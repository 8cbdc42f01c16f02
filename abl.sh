#!/bin/bash
run() { # label, env, flags, extra
  env $2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --flags $3 $4 > gpurun_out/abl_$1.json 2>gpurun_out/abl_$1.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["config"]["steady_state_msamples_per_gpu"], d["roofline"]["kernels_ms_per_step"])
PY
}
run base A=1 0
run nostore A=1 512
run c2048 A=1 0 "--nchan 2048"
run c512 A=1 0 "--nchan 512"
run c256 A=1 0 "--nchan 256"
