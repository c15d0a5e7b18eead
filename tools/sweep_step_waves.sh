#!/bin/bash
# tools/sweep_step_waves.sh : the headline and the d = 64 leg against the resident wavefronts per SIMD of the blocked step kernel
# (rsx_set_option "step_waves", by LDS reservation): the kernel's own duration and the step's period (kernel + the sampler that
# runs in what the kernel leaves free).  Two rounds, the arms alternating.
for round in 1 2; do
for dim in 128 64; do
for w in 5 6 7 8; do
  RSX_STEP_WAVES=$w python3 bench.py --dim $dim --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('d=%-3d step_waves %d  round $round  %8.1f us/step  kernel %8.1f us  %.4g' % ($dim, $w, d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))"
done; done; done
