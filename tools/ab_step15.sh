#!/bin/bash
# tools/ab_step15.sh : TILE = false kernel (independent negatives; configs[3] slice): rounds of wavefronts the batch is cut into; replicas 16 / 32
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 100 --warmup 5 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-20s %-58s value %.3e us/step %.1f kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_rounds2.so recsys_pytorch_amd/build/variants/librsx_rounds4.so recsys_pytorch_amd/build/variants/librsx_rounds8.so; do
one $l "--neg-block 0"; one $l "--users 1250000 --items 1000000 --batch 1250000 --degree 10"; done
one recsys_pytorch_amd/librsx.so "--steps 300"; one recsys_pytorch_amd/librsx.so "--steps 300 --hot-replicas 32"; one recsys_pytorch_amd/librsx.so "--steps 300 --hot 512 --hot-replicas 32"
done
