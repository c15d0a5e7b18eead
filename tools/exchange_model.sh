#!/bin/bash
# tools/exchange_model.sh : the N > 1 schedules on ONE GPU against an exchange of a given length.  A one-rank RCCL group makes the
# collectives identities; the DEVELOPMENT library (librsx_dev.so) holds the trainer's collective stream for RSX_EXCHANGE_DELAY_US
# per full exchange of the item gradients (eight idle workgroups: the footprint of a collective kernel).  Prints ms per step of
# the headline shape for every schedule and delay.  What this cannot show: the HBM / fabric traffic of a real exchange.
export RSX_LIB=$(pwd)/recsys_pytorch_amd/librsx_dev.so RSX_FORCE_SHARDED=1 MASTER_PORT=29641
run() {   # run <label> <env...> -- <bench args>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for d in ${DELAYS:-0 100 200 300 400 600}; do
    env "${envs[@]}" RSX_EXCHANGE_DELAY_US=$d python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-34s delay %4d us  %.1f us/step  (kernel %.1f)' % ('$label', $d, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
}
run "one pass, exchange exposed" RSX_TWO_PASS=0 --
run "two passes (exchange under users)" RSX_TWO_PASS=1 --
run "item ranges x2" RSX_TWO_PASS=0 -- --chunks 2
run "item ranges x3" RSX_TWO_PASS=0 -- --chunks 3
run "one step stale (not synchronous)" RSX_TWO_PASS=0 RSX_STALE_EXCHANGE=1 --
# the same with the stand-in MOVING the message through HBM while it holds the stream (read + write back of the exchanged rows)
run "one pass, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 --
run "two passes, stand-in with traffic" RSX_TWO_PASS=1 RSX_EXCHANGE_TRAFFIC=1 --
run "item ranges x2, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 -- --chunks 2
