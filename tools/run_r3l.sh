mkdir -p gpurun_out/r3l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_post.py -m gpu -q -x -s > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "DEDISP|passed|failed|Error|assert" gpurun_out/r3l/pytest.log | tail -15
