#!/bin/bash
# tools/ab_step.sh lib1 lib2 ... : same-box A/B of the training step between librsx builds (RSX_LIB): headline, iid negatives, d = 64, B = 262144
for round in 1 2; do
for l in "$@"; do
  for args in "" "--neg-block 0" "--dim 64" "--batch 262144"; do
  RSX_LIB=$(pwd)/$l timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 40 --warmup 5 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-22s %-18s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $l)', '$args', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"
  done
done
done
