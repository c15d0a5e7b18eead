#!/bin/bash
# tools/ab_chunks.sh : same box, the headline leg with the step as a pipeline over 0 / 2 / 4 / 8 item ranges
for c in 0 2 4 8 0 4; do
  python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --chunks $c --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('chunks', d['config']['item_chunks'], 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'], 'loss %.4f' % d['config']['mean_bpr_loss'])"
done
