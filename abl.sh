#!/bin/bash
run() { # label, env, flags, extra
  env $2 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --flags $3 $4 > gpurun_out/abl_$1.json 2>gpurun_out/abl_$1.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$1.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels_ms_per_step"]; print("$1", d["value"], d["config"]["steady_state_msamples_per_gpu"], [round(v,3) for v in list(k.values())[:5]])
PY
}
run warm "FRBCH_K1_STAG=12" 0
run s0a "FRBCH_K1_STAG=0" 0
run s12a "FRBCH_K1_STAG=12" 0
run s0b "FRBCH_K1_STAG=0" 0
run s12b "FRBCH_K1_STAG=12" 0
