mkdir -p gpurun_out/r3k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--workload cfg1 --nchan 32 --bw 32 --seconds 2 --steps 1 --warmup 1 --no-cpu --no-host --no-traffic --no-configs"
out=gpurun_out/r3k
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -- python3 bench.py $args > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS --output-format csv -d $out/p2 -- python3 bench.py $args > $out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p3 -- python3 bench.py $args > $out/p3.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/p4 -- python3 bench.py $args > $out/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob('$out/p*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(p)):
        k = row['Kernel_Name']
        if 'k2_lane' not in k and 'k1_wave' not in k: continue
        acc[k][row['Counter_Name']] += float(row['Counter_Value']); n[(k, row['Counter_Name'])] += 1
    for k in acc:
        print(p.split('/')[2], k[:44], {c: round(v / max(1, n[(k, c)])) for c, v in acc[k].items()})
PY
grep -h "k2_lane\|k1_wave" $out/trace/*/*kernel_stats.csv | cut -c1-200
