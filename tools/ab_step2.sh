#!/bin/bash
# tools/ab_step2.sh : occupancy (wavefronts per SIMD the blocked kernel is compiled for) and block size with the pipelined trip loop
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 40 --warmup 5 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-16s %-16s %-36s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['config']['negatives'][:34], d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_w5.so recsys_pytorch_amd/build/variants/librsx_w7.so recsys_pytorch_amd/build/variants/librsx_w8.so; do one $l ""; done
done
for nb in 3 4 5 6 8 10 12 16; do one recsys_pytorch_amd/librsx.so "--neg-block $nb"; done
