#!/bin/bash
# tools/ab_step_small.sh lib1 lib2 ... : same-box A/B of the plain step kernel (small batches) between librsx builds
for round in 1 2; do
for l in "$@"; do
  for b in 4096 16384 65536; do
  RSX_LIB=$(pwd)/$l timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 200 --warmup 10 --batch $b 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-22s B=%-7d value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $l)', $b, d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"
  done
done
done
