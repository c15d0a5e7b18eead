#!/bin/bash
# tools/exchange_model_schedules.sh (round 6): what each exchange schedule costs a step on the HBM SIDE at world 8, measured on ONE GPU with
# the development library, at the BASELINE configs[3] slice (1.25M users x 1M items per rank, d = 128, B = 1.25M: 512 MB of item gradients)
# and at the headline shape (51 MB).  Two ranges per step in every row (bench.py's N > 1 default).
#   A  all-reduce in place + the FULL apply sweep on every rank: the one-rank RCCL group (identities) + the stand-in that holds the
#      collective stream for the wire time and reads / writes the exchanged rows of G once (RSX_EXCHANGE_TRAFFIC=1), then apply(range)
#   C  reduce-scatter -> apply the OWN 1/8 of the rows -> all-gather of the updated rows, as the library's mesh does it (rsx_mesh.hip):
#      rsx_debug_set_mesh_model(8, wire): a one-rank mesh reads its 1/8 slice of G eight times (own + 7 peers; a rank also SERVES 7 reads),
#      applies 1/8 of Q, rewrites the other 7/8 of Q (the gather's writes) and clears G; each phase's kernel is PACED over wire / 2 us (traffic under the wire time)
# wire time of a FULL exchange of S bytes over the 7 links of a rank at L GB/s per link and direction: each phase moves S/8 per link ->
# 2 * (S / 8) / L: S = 512 MB: 2.56 / 1.28 / 0.85 ms at L = 50 / 100 / 150; S = 51.2 MB: 256 / 128 / 85 us.
export RSX_LIB=$(pwd)/recsys_pytorch_amd/librsx_dev.so MASTER_PORT=29653 RSX_FORCE_SHARDED=1 RSX_TWO_PASS=0 RSX_BENCH_MESH_LEG=0
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-64s %7.1f us/step  (kernels %.1f)  chunks %d' % ('$1', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['item_chunks']))"; }
shape() {   # shape <name> <bench args> <wire us per FULL exchange at 50 / 100 / 150 GB/s>
  local name=$1 A=$2; shift 2
  RSX_FORCE_SHARDED=0 python3 bench.py $A 2>/dev/null | line "$name: one GPU, no exchange path"
  for w in "$@"; do
    RSX_EXCHANGE_DELAY_US=$w RSX_EXCHANGE_TRAFFIC=1 python3 bench.py $A --chunks 2 2>/dev/null | line "$name: A all-reduce + full sweep, wire $w us"
    for mb in 0 64; do
      RSX_MESH_BLOCKS=$mb RSX_EXCHANGE=direct RSX_MESH_MODEL_WORLD=8 RSX_MESH_MODEL_DELAY_US=$((w / 2)) python3 bench.py $A --chunks 2 2>/dev/null | line "$name: C mesh RS + own 1/8 applied + AG, wire $w us, mesh_blocks $mb"
    done
  done
}
shape "configs[3] slice" "--users 1250000 --items 1000000 --degree 10 --batch 1250000 --no-legs --score-tiles 0 --no-cpu-baseline --steps 12 --warmup 3" 2560 1280 850 0
shape "headline" "--no-legs --score-tiles 0 --no-cpu-baseline --steps 30 --warmup 5" 256 128 85 0
