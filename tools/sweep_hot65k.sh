#!/bin/bash
# hot-item replica sweep at the base batch (B = 65 536) and the headline (B = 1M)
run() { python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-legs --score-tiles 0 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); c=d['config']; r=d['roofline']
print('B=%d hot=%dx%d: %.1f M triplets/s  step %.1f us kernel %.1f us' % (c['batch_per_gpu'], c['hot_items'], c['hot_replicas'], d['value']/1e6, d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for h in "0 1" "64 16" "256 4" "256 16" "256 64" "1024 16" "4096 8" "4096 32"; do set -- $h; run --batch 65536 --hot $1 --hot-replicas $2; done
for h in "256 16" "1024 16" "64 16"; do set -- $h; run --batch 1000000 --steps 50 --hot $1 --hot-replicas $2; done
