#!/bin/bash
# tools/ab_step6.sh : every leg of the bench line with the item block c = 3 / default and 256- / 512-thread bucketing workgroups
one() { RSX_NEG_BLOCK_EXACT=$2 RSX_LIB=$(pwd)/$1 timeout 600 python3 bench.py --score-tiles 0 --no-cpu-baseline --steps 200 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-18s c=%-2s HEAD value %.3e us/step %.1f kernel %.1f' % ('$(basename $1)', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))
for k,v in d.get('legs',{}).items():
    if isinstance(v,dict) and 'value' in v: print('     %-28s value %.3e us/step %.1f nb %s' % (k, v['value'], v['ms_per_step']*1e3, v.get('neg_block')))
    elif isinstance(v,list): print('     %-28s %s' % (k, ' '.join('%d:%.3e' % (x['batch_per_gpu'], x['value']) for x in v)))
"; }
for round in 1 2; do
one recsys_pytorch_amd/librsx.so ""
one recsys_pytorch_amd/librsx.so 3
one recsys_pytorch_amd/build/variants/librsx_ct256.so 3
one recsys_pytorch_amd/build/variants/librsx_ct256.so 2
done
