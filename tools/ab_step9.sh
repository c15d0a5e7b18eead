#!/bin/bash
# tools/ab_step9.sh : what the sampler costs the loop -- development library, the loop with and without the sampler kernels beside it
one() { RSX_SAMPLER_REPLAY=$1 RSX_LIB=$(pwd)/recsys_pytorch_amd/librsx_dev.so timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('replay=%s %-12s value %.3e  us/step %.1f  kernel %.1f' % ('$1', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do for r in 0 1; do one $r ""; one $r "--chunks 2"; one $r "--batch 65536"; done; done
