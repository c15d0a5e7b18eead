#!/bin/bash
# development sweep: batch size, blocked negatives, hot-item replicas on one MI355X
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --score-tiles 0 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); c=d['config']; r=d['roofline']
print('B=%d %s neg=%s hot=%dx%d: %.0f M triplets/s  step %.1f us kernel %.1f us frac %.3f' % (c['batch_per_gpu'], c['item_popularity'], c['negatives'][:14], c['hot_items'], c['hot_replicas'], d['value']/1e6, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['frac']))"; }
for B in 200000 300000 500000 1000000; do
  for nb in 0 4 8 16; do run --batch $B --neg-block $nb; done
done
run --batch 1000000 --neg-block 8 --popularity uniform --hot 0
run --batch 1000000 --neg-block 0 --popularity uniform --hot 0
run --batch 1000000 --neg-block 8 --hot 1024 --hot-replicas 16
run --batch 1000000 --neg-block 8 --dim 64
