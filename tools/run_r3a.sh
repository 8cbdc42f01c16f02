mkdir -p gpurun_out/r3a
./tools/micro/cumask_probe 160 > gpurun_out/r3a/probe.txt 2>&1; cat gpurun_out/r3a/probe.txt
./tools/micro/cumask_probe 96 >> gpurun_out/r3a/probe.txt 2>&1
bash tools/overlap_sweep.sh r3a cfg2 "1 0 128 160 192 $((160 | 2<<16)) $((160 | 8<<16)) $((144 | 4<<16)) $((176 | 4<<16))" 2>&1 | tee gpurun_out/r3a/sweep_cfg2.txt
bash tools/overlap_sweep.sh r3a cfg3 "1 0 96 128 144 $((112 | 1<<16)) $((112 | 2<<16))" 2>&1 | tee gpurun_out/r3a/sweep_cfg3.txt
