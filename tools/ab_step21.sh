#!/bin/bash
# tools/ab_step21.sh : the dense form of the apply (Q requested with G) from batch * factor >= items on; 400 steps per line
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 400 --warmup 20 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-18s %-16s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2 3; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_ad2.so recsys_pytorch_amd/build/variants/librsx_ad8.so; do one $l "--batch 65536"; one $l "--batch 16384"; done
done
