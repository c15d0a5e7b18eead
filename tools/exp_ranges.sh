#!/bin/bash
# tools/exp_ranges.sh : the range pipelines on one GPU, no exchange: stream priorities per range (0 = highest .. 2 = lowest)
one() {  # one <chunks> <prio string or ->
  ( [ "$2" != "-" ] && export RSX_RANGE_PRIO=$2; timeout 120 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --chunks $1 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('chunks', d['config']['item_chunks'], 'prio', os.environ.get('RSX_RANGE_PRIO','default'), 'ms/step %.4f' % d['ms_per_step'], 'span_ms %.4f' % r['kernel_ms'], 'loss %.4f' % d['config']['mean_bpr_loss'])" || echo "chunks $1 prio $2 FAILED/timeout" )
}
one 4 -; one 4 0112; one 4 0111; one 4 1111; one 4 0011; one 4 0122
one 3 -; one 3 011; one 3 111; one 3 001
one 2 -; one 2 11; one 2 00; one 2 12
one 8 01111111; one 6 011111
