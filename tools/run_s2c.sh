cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2c
plain="--no-cpu --no-host --no-traffic --no-configs --no-steady"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2c/trace -- python3 bench.py --workload cfg3 --overlap 33554624 $plain --steps 2 --warmup 1 > gpurun_out/s2c/trace_bench.json 2> gpurun_out/s2c/trace.err
echo "trace rc=$?"
find gpurun_out/s2c/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/s2c/kernel_trace.csv \;
rm -rf gpurun_out/s2c/trace
export FRBCH_LIB=$GRAFT_REPO_ROOT/frb_baseband_amd/csrc/libfrbch_exp.so
for w in 1 2 3 4 6 8 16; do
FRBCH_QUANT_WGS=$w timeout -k 10 300 python3 bench.py --workload cfg3 --no-cpu --no-traffic --no-configs --no-host --steps 6 --warmup 2 > gpurun_out/s2c/b.json 2> gpurun_out/s2c/b.err
python3 -c "
import json; d=json.loads(open('gpurun_out/s2c/b.json').read().strip().splitlines()[-1]); print('wgs/CU $w', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done
