#!/bin/bash
# usage: tools/pmc.sh <tag> [bench args...]   -- PMC passes for the hot kernels (run on the GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
mkdir -p $out
args="--steps 1 --warmup 1 --seconds 2 --no-cpu --no-host --no-traffic $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -- python3 bench.py $args > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS --output-format csv -d $out/p2 -- python3 bench.py $args > $out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p3 -- python3 bench.py $args > $out/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/p4 -- python3 bench.py $args > $out/p4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p5 -- python3 bench.py $args > $out/p5.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob('$out/p*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(p)):
        k = row['Kernel_Name']
        if 'frbch' not in k: continue
        acc[k][row['Counter_Name']] += float(row['Counter_Value']); 
    for k in acc:
        print(p.split('/')[2], k[:40], {c: round(v) for c, v in acc[k].items()})
PY
