#!/bin/bash
run() { # label, env, flags, extra
  env $2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --flags $3 $4 > gpurun_out/abl_$1.json 2>gpurun_out/abl_$1.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["config"]["steady_state_msamples_per_gpu"], d["roofline"]["kernels_ms_per_step"])
PY
}
run k1half_s0 FRBCH_K1_STAG=0 192
run k1half_s3 FRBCH_K1_STAG=3 192
run k1half_s6 FRBCH_K1_STAG=6 192
run k1two_s3 FRBCH_K1_STAG=3 64
