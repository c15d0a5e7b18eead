#!/bin/bash
# tools/exchange_model_config3.sh : the BASELINE configs[3] one-rank slice (1.25M users x 1M items, d = 128, B = 1.25M < 2 I) under every
# N > 1 schedule on ONE GPU against an exchange of a given length (one-rank RCCL group: the collectives are identities; the
# DEVELOPMENT library holds the trainer's collective stream for RSX_EXCHANGE_DELAY_US per full exchange of the 512 MB of item
# gradients).  What this cannot show: the HBM / fabric traffic of a real exchange.
export RSX_LIB=$(pwd)/recsys_pytorch_amd/librsx_dev.so MASTER_PORT=29643
A="--users 1250000 --items 1000000 --degree 10 --batch 1250000 --no-legs --score-tiles 0 --no-cpu-baseline --steps 12 --warmup 3"
run() {   # run <label> <env...> -- <bench args>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for d in ${DELAYS:-0 500 1000 1500}; do
    env "${envs[@]}" RSX_EXCHANGE_DELAY_US=$d python3 bench.py $A "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-52s delay %4d us  %7.1f us/step  (kernels %.1f)  chunks %d' % ('$label', $d, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['item_chunks']))"
  done
}
DELAYS=0 run "one GPU, no exchange path" --
DELAYS=0 run "one GPU, item ranges x2, no exchange path" -- --chunks 2
export RSX_FORCE_SHARDED=1
run "one pass, exchange exposed" RSX_TWO_PASS=0 --
run "two passes (exchange under users)" RSX_TWO_PASS=1 --
run "item ranges x2, applies on the collective stream" RSX_TWO_PASS=0 RSX_APPLY_STREAM=0 -- --chunks 2
run "item ranges x2, applies on their own stream" RSX_TWO_PASS=0 RSX_APPLY_STREAM=1 -- --chunks 2
run "item ranges x3, own apply stream" RSX_TWO_PASS=0 RSX_APPLY_STREAM=1 -- --chunks 3
run "item ranges x4, own apply stream" RSX_TWO_PASS=0 RSX_APPLY_STREAM=1 -- --chunks 4
# the same with the stand-in MOVING the message through HBM while it holds the stream
run "one pass, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 --
run "item ranges x2, stand-in with traffic" RSX_TWO_PASS=0 RSX_APPLY_STREAM=0 RSX_EXCHANGE_TRAFFIC=1 -- --chunks 2
