timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4096" 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu --steps 2 --warmup 1 --nchan 4096 --bw 64 --seconds 2.2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
