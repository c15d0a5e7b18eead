#!/bin/bash
# tools/ab_step7.sh : item ranges beside the new block-size rule
one() { RSX_NEG_BLOCK_EXACT=$2 timeout 600 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 --chunks $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('chunks %s c=%-2s value %.3e us/step %.1f kernel %.1f  %s' % ('$1', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['config']['negatives'][:40]))"; }
for round in 1 2; do for ch in 0 2 3; do for c in "" 3 6; do one $ch $c; done; done; done
