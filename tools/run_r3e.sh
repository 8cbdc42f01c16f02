mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "bench rc=$?"
cut -c1-3000 gpurun_out/r3e/bench.json
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3e/pytest.log
