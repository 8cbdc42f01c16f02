#!/bin/bash
# All BASELINE configurations in one go (run on the GPU box): short benches, one summary line each -> gpurun_out/sweep_<tag>.log
tag=${1:-configs}
bash tools/sweep.sh $tag \
  "cfg1_128ch|--nchan 128 --bw 16" \
  "cfg2_1024ch|" \
  "cfg3_1024ch_d4|--pol 4" \
  "cfg4_4096ch|--nchan 4096 --bw 64 --seconds 5" \
  "cfg4_4096ch_t8|--nchan 4096 --bw 64 --seconds 5 --tscrunch 8" \
  "cfg5_2048ch_coherent|--nchan 2048 --freq-res 4096 --dm 56.7 --coherent --freq 1400" \
  "c2048|--nchan 2048" 2>&1 | tee gpurun_out/sweep_$tag.log
