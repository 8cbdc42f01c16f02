cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2g
timeout -k 10 600 python3 -m pytest tests/test_gpu_pins.py -m gpu -x -q -k "lanes_give" > gpurun_out/s2g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2g/pytest.log
bash tools/overlap_sweep.sh s2g cfg3 "1 50331808 50331824 50331840 50331856 50331872"
plain="--no-cpu --no-host --no-traffic --no-configs --no-steady"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2g/trace -- python3 bench.py --workload cfg3 --overlap 50331824 $plain --steps 2 --warmup 1 > gpurun_out/s2g/trace_bench.json 2> gpurun_out/s2g/trace.err
echo "trace rc=$?"
find gpurun_out/s2g/trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/s2g/kernel_trace.csv \;
rm -rf gpurun_out/s2g/trace
