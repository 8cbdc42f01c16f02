python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --share-gpu --seconds 4 2>&1 | tail -2 | cut -c1-900
