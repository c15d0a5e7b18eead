#!/bin/bash
# tools/ab_step4.sh : item block size c of the blocked negatives x workgroup size of the bucketing kernel (300 steps per line)
one() { RSX_NEG_BLOCK_EXACT=$2 RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-22s c=%s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for c in 2 3 4 5 6 8; do
for l in recsys_pytorch_amd/build/variants/librsx_ct512.so recsys_pytorch_amd/build/variants/librsx_ct256.so; do one $l $c; done
done; done
