mkdir -p gpurun_out/r3d
export FRBCH_LIB=$PWD/frb_baseband_amd/csrc/libfrbch_exp.so
M2=$((2<<24))
for w in 1 2 4 8 32; do
  echo "== FRBCH_QUANT_WGS=$w"
  FRBCH_QUANT_WGS=$w bash tools/overlap_sweep.sh r3d cfg3 "1 $((192|M2)) $((176|M2))" 2>&1
done | tee gpurun_out/r3d/sweep_cfg3.txt
FRBCH_QUANT_WGS=2 bash tools/overlap_sweep.sh r3d cfg2 "1" 2>&1 | tee gpurun_out/r3d/sweep_cfg2.txt
FRBCH_QUANT_WGS=8 bash tools/overlap_sweep.sh r3d cfg2 "1" 2>&1 | tee -a gpurun_out/r3d/sweep_cfg2.txt
