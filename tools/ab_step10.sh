#!/bin/bash
# tools/ab_step10.sh : positions per lane group and trip at d = 64 / 32 (RSX_STEP_NP_SMALL 2 / 3 / 4), 300 steps per line
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 --dim $2 $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-16s d=%-3s %-14s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', '$3', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for l in recsys_pytorch_amd/build/variants/librsx_np2.so recsys_pytorch_amd/build/variants/librsx_np3.so recsys_pytorch_amd/librsx.so; do
one $l 64 ""; one $l 32 ""; one $l 64 "--neg-block 0"; done; done
