cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2h
export FRBCH_LIB=$GRAFT_REPO_ROOT/frb_baseband_amd/csrc/libfrbch_exp.so
for v in 0 1 2; do
echo "== FRBCH_QUANT_EXCL=$v maxb default"
FRBCH_QUANT_EXCL=$v bash tools/overlap_sweep.sh s2h cfg3 "50331792 50331808 50331824"
echo "== FRBCH_QUANT_EXCL=$v maxb 152"
FRBCH_QUANT_EXCL=$v bash tools/overlap_sweep.sh s2h cfg3 "50331776 50331792 50331808 50331824" --maxb 152
done
echo "== no overlap maxb 152"
bash tools/overlap_sweep.sh s2h cfg3 "1" --maxb 152
