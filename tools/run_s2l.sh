cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2l
for rep in 1 2; do
bash tools/overlap_sweep.sh s2l cfg3 "0 50331816 50331824 50331832 50331840"
done
