mkdir -p gpurun_out/r3g
export FRBCH_LIB=$PWD/frb_baseband_amd/csrc/libfrbch_exp.so FRBCH_TIMING=1
timeout -k 10 400 python3 tools/scan_timing.py 8 5 10 > gpurun_out/r3g/scan8_iquv.txt 2>&1; tail -60 gpurun_out/r3g/scan8_iquv.txt
timeout -k 10 400 python3 tools/scan_timing.py 8 2 10 > gpurun_out/r3g/scan8_I.txt 2>&1; tail -40 gpurun_out/r3g/scan8_I.txt
