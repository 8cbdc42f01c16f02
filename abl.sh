#!/bin/bash
run() { # label, env, flags, extra
  env $2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --flags $3 $4 > gpurun_out/abl_$1.json 2>gpurun_out/abl_$1.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$1.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels_ms_per_step"]; print("$1", d["value"], d["config"]["steady_state_msamples_per_gpu"], [round(v,3) for v in list(k.values())[:5]])
PY
}
for p in 16 48 96 160 272 400 1040 1296 4112 8208; do run pad$p FRBCH_SPILL_PAD=$p 0; done
