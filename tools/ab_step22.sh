#!/bin/bash
# tools/ab_step22.sh : positions per bucketing workgroup of the ordering sampler (4096 / 2048 / 1024: 244 / 488 / 977 workgroups at B = 1M); 300 steps per line
one() { RSX_LIB=$(pwd)/$1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-18s %-16s value %.3e  us/step %.1f  kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2 3; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_cp2048.so recsys_pytorch_amd/build/variants/librsx_cp1024.so; do one $l ""; one $l "--chunks 2"; one $l "--batch 262144"; done
done
for l in cp2048 cp1024; do RSX_LIB=$(pwd)/recsys_pytorch_amd/build/variants/librsx_$l.so python -m pytest tests -x -q -m gpu -k "sampl or chunk" 2>&1 | tail -1; done
