#!/bin/bash
# tools/ab_step16.sh : replicas of the popular rows at the headline and the base batch; the row-access microbenchmark with a log-structured table
one() { timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-48s value %.3e us/step %.1f kernel %.1f' % ('$1', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for h in "--hot-replicas 16" "--hot-replicas 32" "--hot-replicas 64" "--hot 128 --hot-replicas 64" "--batch 65536 --hot-replicas 16" "--batch 65536 --hot-replicas 32" "--batch 65536 --hot-replicas 64" "--dim 64 --hot-replicas 16" "--dim 64 --hot-replicas 32"; do one "$h"; done; done
./tools/atomic_bench 1000000 1000000
