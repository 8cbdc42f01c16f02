timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 120 python bench.py --no-cpu --steps 5 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['steady_state_msamples_per_gpu'], d['roofline']['kernels_ms_per_step'])"
