#!/bin/bash
# smoke-test bench.py's N>1 code path on ONE GPU: two ranks share it, gloo moves G (not a perf number)
export RSX_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for parts in 0 1; do
  RSX_TWO_PASS=$parts timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 \
    bench.py --gpus 2 --steps 6 --warmup 2 --users 200000 --batch 200000 --items 50000 --score-tiles 0 2>&1 | tail -1 | cut -c1-400
done
