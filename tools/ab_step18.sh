#!/bin/bash
# tools/ab_step18.sh : resident wavefronts per SIMD the blocked kernel is compiled for (5 / 6 / 7) at blocks of 2; grid cap of the LightGCN product
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-18s %-12s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_w5.so recsys_pytorch_amd/build/variants/librsx_w7.so; do one $l ""; one $l "--dim 64"; done
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_spg4.so recsys_pytorch_amd/build/variants/librsx_spg16.so recsys_pytorch_amd/build/variants/librsx_spg64.so; do echo -n "$(basename $l): "; RSX_LIB=$(pwd)/$l timeout 600 python3 tools/bench_lightgcn.py 2>/dev/null | grep -E "spmm:|train_step" | tr '\n' ' '; echo; done
done
