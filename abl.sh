#!/bin/bash
run() { # label, extra
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu $2 > gpurun_out/cfg_$1.json 2>gpurun_out/cfg_$1.err || { echo "$1 FAILED"; tail -n 3 gpurun_out/cfg_$1.err; return; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/cfg_$1.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels_ms_per_step"]; print("$1", round(d["value"]), round(d["config"]["steady_state_msamples_per_gpu"]), d["ms_per_step"], {a:round(b,3) for a,b in k.items() if b>0})
PY
}
run c1024_I ""
run c1024_d4 "--pol 4"
run c256 "--nchan 256 --bw 16"
run c512 "--nchan 512"
run c2048 "--nchan 2048 --bw 64"
run c2048_d4 "--nchan 2048 --bw 64 --pol 4"
run c4096 "--nchan 4096 --bw 64"
run cfg5 "--coherent --dm 56.7 --nchan 2048 --freq-res 4096 --freq 1400"
run c128 "--nchan 128 --bw 16"
