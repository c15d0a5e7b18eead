#!/bin/bash
# tools/ab_step24.sh : item ranges with the last range on the caller's stream (one HIP stream fewer) vs a stream per range; 300 steps per line
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-16s %-12s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2 3; do for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_rrs.so; do one $l "--chunks 2"; one $l "--chunks 3"; one $l "--chunks 4"; done; done
RSX_LIB=$(pwd)/recsys_pytorch_amd/build/variants/librsx_rrs.so python -m pytest tests -x -q -m gpu -k "chunk or rccl_with_one_rank" 2>&1 | tail -2
