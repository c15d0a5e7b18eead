#!/bin/bash
# tools/ab_step23.sh : the samplers of steps t+1 and t+2 on TWO lowest-priority streams (a sampler workspace that holds two) vs one; 300 steps per line
one() { RSX_TWO_SAMPLERS=$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('two_samplers=%s %-16s value %.3e  us/step %.1f  kernel %.1f' % ('$1', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2 3; do for t in 0 1; do one $t ""; one $t "--chunks 2"; one $t "--chunks 3"; one $t "--batch 262144"; one $t "--neg-block 0"; one $t "--dim 64"; done; done
RSX_TWO_SAMPLERS=1 python -m pytest tests -x -q -m gpu -k "trainer or native or chunk or full_size or sampl" 2>&1 | tail -2
