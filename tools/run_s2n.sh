cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2n
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/s2n/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s2n/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
