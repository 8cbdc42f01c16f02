#!/bin/bash
# usage: tools/overlap_sweep.sh <tag> <workload> "<overlap values>" [bench args]   (run on the GPU box)
# one short bench per frbch_config.overlap value (front-lane CUs | batches << 16; 1 = no overlap), one summary line each
tag=$1; wl=$2; vals=$3; shift 3
mkdir -p gpurun_out/$tag
for v in $vals; do
  timeout -k 10 240 python3 bench.py --workload $wl --no-cpu --no-host --no-traffic --no-configs --steps 8 --warmup 3 --overlap $v "$@" \
    > gpurun_out/$tag/${wl}_$v.json 2> gpurun_out/$tag/${wl}_$v.err || { echo "$wl overlap=$v FAILED"; tail -3 gpurun_out/$tag/${wl}_$v.err; continue; }
  python3 - "$wl" "$v" gpurun_out/$tag/${wl}_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "overlap", sys.argv[2], "front", int(sys.argv[2]) & 0xFFFF, "batches", int(sys.argv[2]) >> 16, "value", round(d["value"]), "steady", d["config"]["steady_state_msamples_per_gpu"],
      "ms/step", d["ms_per_step"], "whole", r["whole_path"]["frac"], r["kernels_ms_per_step"])
PY
done
