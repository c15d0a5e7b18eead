#!/bin/bash
# tools/ab_queues.sh : item ranges against the number of hardware queues the HIP runtime multiplexes the streams onto (GPU_MAX_HW_QUEUES, default 4)
one() { env $1 timeout 300 python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 300 --warmup 10 --chunks $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-24s chunks %s value %.3e us/step %.1f kernel %.1f' % ('$1', '$2', d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3))"; }
for round in 1 2; do for q in "X=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2"; do for c in 0 2 3 4; do one $q $c; done; done; done
