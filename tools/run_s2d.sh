cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2d
timeout -k 10 200 python3 tools/enqueue_time.py --workload cfg3 2>&1 | grep -v Warning
timeout -k 10 200 python3 tools/enqueue_time.py --workload cfg3 --overlap 33554624 2>&1 | grep -v Warning
timeout -k 10 200 python3 tools/enqueue_time.py --workload cfg2 2>&1 | grep -v Warning
plain="--no-cpu --no-host --no-traffic --no-configs --no-steady"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2d/trace -- python3 bench.py --workload cfg3 $plain --steps 2 --warmup 1 > gpurun_out/s2d/trace_bench.json 2> gpurun_out/s2d/trace.err
echo "trace rc=$?"
find gpurun_out/s2d/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/s2d/kernel_trace.csv \;
rm -rf gpurun_out/s2d/trace
