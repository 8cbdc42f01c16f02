mkdir -p gpurun_out/r3c
M2=$((2<<24))
bash tools/overlap_sweep.sh r3c cfg3 "1 0 $((208|M2)) $((176|M2)) $((160|M2)) $((224|M2))" 2>&1 | tee gpurun_out/r3c/sweep_cfg3.txt
bash tools/overlap_sweep.sh r3c cfg2 "1" 2>&1 | tee gpurun_out/r3c/sweep_cfg2.txt
bash tools/overlap_sweep.sh r3c cfg3 "1 $((192|M2)) $((208|M2))" --pol 2 2>&1 | tee gpurun_out/r3c/sweep_cfg3_stokesI.txt
