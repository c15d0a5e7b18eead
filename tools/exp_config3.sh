#!/bin/bash
# tools/exp_config3.sh : the configs[3] one-rank slice (1.25M users x 1M items, B = 1.25M < 2 I) through the item ranges
# (blocked negatives forced on: RSX_BLOCKED_ANY_BATCH=1) against its default ordered layout (TILE = false kernel)
A="--users 1250000 --items 1000000 --degree 10 --batch 1250000 --no-legs --score-tiles 0 --no-cpu-baseline --steps 12 --warmup 2"
one() { ( export $1; shift; python3 bench.py $A "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-40s' % os.environ.get('LABEL',''), 'chunks', d['config']['item_chunks'], d['config']['negatives'][:34], '| ms/step %.4f' % d['ms_per_step'], 'kernel/span %.4f' % r['kernel_ms'], r['kernel'])" ) }
one LABEL=default --chunks 0
for nb in 8 16; do
  one "LABEL=blocked_nb$nb RSX_BLOCKED_ANY_BATCH=1" --chunks 0 --neg-block $nb
  one "LABEL=ranges2_nb$nb RSX_BLOCKED_ANY_BATCH=1" --chunks 2 --neg-block $nb
  one "LABEL=ranges3_nb$nb RSX_BLOCKED_ANY_BATCH=1" --chunks 3 --neg-block $nb
done
