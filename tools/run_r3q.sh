cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3q/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3q/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3q/smoke.log
bash tools/profile.sh r03_final > gpurun_out/r3q/profile.log 2>&1; tail -30 gpurun_out/r3q/profile.log | cut -c1-600
