#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/rx && cp -r include frb_baseband_amd /tmp/rx/ && make -C /tmp/rx/frb_baseband_amd/csrc clean >/dev/null && make -C /tmp/rx/frb_baseband_amd/csrc EXPERIMENTS=1 -j16 > gpurun_out/abl4_build.log 2>&1
cp frb_baseband_amd/csrc/libfrbch.so /tmp/libfrbch_product.so
cp /tmp/rx/frb_baseband_amd/csrc/libfrbch.so frb_baseband_amd/csrc/libfrbch.so || exit 1
ARGS='--nchan 4096 --bw 64 --seconds 5' bash tools_ablate.sh cfg4 0 4096 16384 8192 20480 512 1024 1536 65536 > gpurun_out/abl4w.txt 2>&1
cp /tmp/libfrbch_product.so frb_baseband_amd/csrc/libfrbch.so
cat gpurun_out/abl4w.txt
