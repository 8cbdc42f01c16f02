cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3r
bash tools/profile.sh r03_final > gpurun_out/r3r/profile.log 2>&1; tail -12 gpurun_out/r3r/profile.log | cut -c1-300
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "full_length" > gpurun_out/r3r/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3r/pytest.log
