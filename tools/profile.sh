#!/bin/bash
# Round artifacts (run on the GPU box): the default bench line (with its own live PMC traffic passes, the host-inclusive legs and
# the other configurations), rocprofv3 kernel stats of the same workload, per-kernel HBM traffic from separate PMC passes.
tag=${1:-r04_final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 700 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done rc=$?"
plain="--no-cpu --no-host --no-traffic --no-configs --no-steady"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $plain > $out/trace_bench.json 2> $out/trace.err
echo "trace done rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $plain --steps 1 --warmup 1 > $out/fetch.log 2>&1
echo "fetch done rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $plain --steps 1 --warmup 1 > $out/write.log 2>&1
echo "write done rc=$?"
python3 - <<PY
import csv, glob, collections, json
res = {}
for name, cnt in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for p in glob.glob("$out/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] != cnt: continue
            k = row["Kernel_Name"]
            acc[k] += float(row["Counter_Value"]); n[k] += 1
        for k in acc:
            if "frbch" in k:
                res.setdefault(k, {})[cnt + "_KB_total"] = acc[k]; res[k]["dispatches"] = n[k]
json.dump(res, open("$out/pmc_traffic.json", "w"), indent=1)
stats = glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    import shutil
    shutil.copy(stats[0], "$out/kernel_stats.csv")
    print(open(stats[0]).read()[:2500])
PY
cut -c1-1500 $out/bench.json
