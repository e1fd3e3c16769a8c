#!/bin/bash
# exercise bench.py's N>1 flow (sharding, gather, host merge, max-over-ranks timing) with 2 and 4 ranks sharing the one GPU
export TMPDIR=/tmp
for n in 2 4; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 2 --warmup 1 --gbases 1 --backend gloo --share-gpu 2>&1 | tail -3
done
python bench.py --steps 2 --warmup 1 --gbases 1 --no-cpu-baseline 2>&1 | tail -1
# merged sketch of N ranks x 1 Gbase must equal the 1-rank sketch of N Gbase (same read indices)
python bench.py --steps 1 --warmup 0 --gbases 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['sketch_check'])"
