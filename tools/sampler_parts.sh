#!/bin/bash
# tools/sampler_parts.sh [dim ...] : which part of the sampler costs the step what?  (GPU box; DEVELOPMENT library, timing only.)  The loop
# steps on replayed batches -- an identical step workload in every arm -- while the sampler runs beside it into a shadow buffer
# (rsx_debug_set_sampler_replay(2)) with parts of it switched off (rsx_debug_set_sample_ablation: csrc/rsx_sample.hip).
lib=$PWD/recsys_pytorch_amd/librsx_dev.so
for dim in ${@:-128 64}; do
  for arm in "1 0:no sampler at all" "2 0:the whole sampler" "2 2:bucketing pass only" "2 1:sort pass only (stale chunks)" "2 4:whole, no LDS sort" \
             "2 8:whole, no rejection reads" "2 12:whole, no LDS sort, no rejection reads" "2 28:whole, sort pass empty (no gather either)" "2 3:nothing launched (host calls only)" "2 40:MOCK: negative drawn in the bucketing pass (row in registers, 4th word per pair), no rejection reads in the sort pass" "2 44:MOCK ... and no LDS sort"; do
    set -- ${arm%%:*}; name=${arm#*:}
    RSX_LIB=$lib RSX_SAMPLER_REPLAY=$1 RSX_SAMPLE_ABLATION=$2 python bench.py --dim $dim --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('d=%-3s %-60s %8.1f us/step  kernel %8.1f us' % ('$dim', '$name', d['ms_per_step']*1e3, r['kernel_ms']*1e3))"
  done
done
