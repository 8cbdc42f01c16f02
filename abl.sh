timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for f in 0; do echo "flags=$f"; timeout -k 10 120 python bench.py --no-cpu --steps 5 --flags $f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"; done
