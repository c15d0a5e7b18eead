#!/bin/bash
# tools/ab_step5.sh : CUs of its own for the sampler (RSX_CU_SPLIT) x workgroup size of the bucketing kernel; LDS-floor occupancy caps
one() { RSX_CU_SPLIT=$2 RSX_LIB=$(pwd)/recsys_pytorch_amd/build/variants/librsx_$1.so timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-10s split=%-3s value %.3e  us/step %.1f  kernel %.1f' % ('$1', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for l in cs1024 cs512 cs256; do for sp in 0 16 32 64; do one $l $sp; done; done
for l in l5ct256 l5ct256u l4ct256 l5ct512 l4ct512; do one $l 0; done
done
