#!/bin/bash
# tools/ab_step11.sh : popular-row replicas beside blocks of 2 (hot items x replicas), and the block size at B = 262 144; 300 steps per line
one() { env $1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-22s %-34s value %.3e us/step %.1f kernel %.1f' % ('$1', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do
for h in "--hot 256 --hot-replicas 16" "--hot 0" "--hot 64 --hot-replicas 16" "--hot 1024 --hot-replicas 16" "--hot 256 --hot-replicas 8" "--hot 256 --hot-replicas 32" "--hot 1024 --hot-replicas 8" "--hot 4096 --hot-replicas 4"; do one X=1 "$h"; done
for c in 3 4 5 6 8; do one RSX_NEG_BLOCK_EXACT=$c "--batch 262144"; done
done
