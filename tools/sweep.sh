#!/bin/bash
# usage: tools/sweep.sh <tag> "<label>|<bench args>" ...   -- short bench runs (run on the GPU box), one summary line each
tag=$1; shift
out=gpurun_out/sweep_$tag; mkdir -p $out
for spec in "$@"; do
  label=${spec%%|*}; args=${spec#*|}; envs=""
  if [[ "$args" == *";"* ]]; then envs=${args%%;*}; args=${args#*;}; fi
  env $envs timeout -k 10 240 python3 bench.py --no-cpu --no-host --no-traffic --steps 5 --warmup 3 $args > $out/$label.json 2> $out/$label.err || { echo "$label FAILED"; tail -3 $out/$label.err; exit 1; }
  python3 - "$label" "$out/$label.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]; k = r["kernels_ms_per_step"]
print(sys.argv[1], "value", round(d["value"]), "steady", round(d["config"]["steady_state_msamples_per_gpu"]), "ms/step", d["ms_per_step"],
      "dom", r["kernel"], "frac", r["frac"], {a: b for a, b in k.items() if b > 0}, flush=True)
PY
done
