#!/bin/bash
# tools/ab_step12.sh : same-box check of the generalised trip loop (NP as a template constant) against the loop before it
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-22s %-12s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2 3; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_nohs.so; do one $l ""; one $l "--dim 64"; one $l "--batch 262144"; one $l "--chunks 2"; done
done
