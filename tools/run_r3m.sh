mkdir -p gpurun_out/r3m
export FRBCH_LIB=$PWD/frb_baseband_amd/csrc/libfrbch_exp.so
M2=$((2<<24))
for w in 1; do
  echo "== FRBCH_QUANT_WGS=$w"
  FRBCH_QUANT_WGS=$w bash tools/overlap_sweep.sh r3m cfg3 "1 $((256|M2))" 2>&1
done | tee gpurun_out/r3m/sweep_cfg3_w1.txt
