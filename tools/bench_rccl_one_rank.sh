#!/bin/bash
# the N>1 step schedules over RCCL with a process group of ONE rank (all a one-GPU box offers): what the exchange
# path costs before any byte crosses xGMI -- the callbacks into the interpreter, the collective's own stream and
# the two-pass split.  Usage (on the GPU box): bash tools/bench_rccl_one_rank.sh > gpurun_out/rccl1.txt
cd "$(dirname "$0")/.."
for cfg in "0 allreduce 0 0" "1 allreduce 0 0" "1 allreduce 1 0" "1 scatter_gather 0 0" "1 scatter_gather 1 0" "1 allreduce 0 1" "1 scatter_gather 0 1"; do
  set -- $cfg
  RSX_FORCE_SHARDED=$1 RSX_EXCHANGE=$2 RSX_TWO_PASS=$3 RSX_STALE_EXCHANGE=$4 python bench.py --no-legs --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null |
    python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('forced=$1 exchange=$2 two_pass=$3 stale=$4  ms_per_step %.4f  value %.4g  kernel_ms %.4f  parallelism: %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['config']['parallelism']))"
done
