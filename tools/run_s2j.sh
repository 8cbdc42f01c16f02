cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2j
export FRBCH_LIB=$GRAFT_REPO_ROOT/frb_baseband_amd/csrc/libfrbch_exp.so
for f in 0 4096 8192 16384 12288 20480 24576 28672; do
timeout -k 10 300 python3 bench.py --workload cfg3 --nif 2 --overlap 1 --no-cpu --no-traffic --no-configs --no-host --steps 6 --warmup 2 --flags $f > gpurun_out/s2j/b.json 2> gpurun_out/s2j/b.err
python3 -c "
import json,sys; d=json.loads(open('gpurun_out/s2j/b.json').read().strip().splitlines()[-1]); ss=d['config'].get('steady_state_msamples_per_gpu'); 
print('flags', sys.argv[1], 'value', d['value'], 'steady', ss, 'steady ms per IF %.3f' % (640e6/ (ss*1e6)*1e3), d['roofline']['kernels_ms_per_step'])" "$f"
done
