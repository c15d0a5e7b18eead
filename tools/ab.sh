#!/bin/bash
# tools/ab.sh [-r rounds] [-a "<bench.py args>"] variant[:-Dflag[,-Dflag...]] ... : same-box A/B of librsx builds on bench.py.
# A variant named `tree` is the library in the tree (recsys_pytorch_amd/librsx.so); any other name is built HERE from the current
# csrc/ with the given flags through tools/build_variant.py (into recsys_pytorch_amd/build/variants/, which travels with gpurun: build
# the variants in the CPU container first -- `tools/ab.sh -b ...` only builds) and loaded through RSX_LIB.  The arms alternate inside
# every round; one line per run: ms per step, the step kernel's mean, the value.
#   tools/ab.sh -b old:-DRSX_FAST_LOSS=0                 (CPU container: build)
#   tools/ab.sh -r 3 tree old:-DRSX_FAST_LOSS=0          (GPU box: measure; default args: the headline alone)
#   tools/ab.sh -a "--dim 64 --no-legs --score-tiles 0 --no-cpu-baseline" tree old
rounds=2; args="--no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5"; build_only=0
while getopts "r:a:b" o; do case $o in r) rounds=$OPTARG;; a) args=$OPTARG;; b) build_only=1;; esac; done
shift $((OPTIND - 1))
root=$(cd "$(dirname "$0")/.." && pwd)
libs=(); names=()
for spec in "$@"; do
  name=${spec%%:*}; flags=""; [ "$spec" != "$name" ] && flags=${spec#*:}
  if [ "$name" = "tree" ]; then lib=$root/recsys_pytorch_amd/librsx.so
  else
    lib=$root/recsys_pytorch_amd/build/variants/librsx_$name.so
    if [ $build_only = 1 ] || [ ! -f "$lib" ]; then python3 "$root/tools/build_variant.py" "$name" ${flags//,/ } > /dev/null || exit 1; fi
  fi
  libs+=("$lib"); names+=("$name")
done
[ $build_only = 1 ] && { ls -la "${libs[@]}"; exit 0; }
for round in $(seq 1 $rounds); do
  for q in "${!libs[@]}"; do
    RSX_LIB=${libs[$q]} timeout 300 python3 "$root/bench.py" $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-14s round $round  %8.1f us/step  kernel %8.1f us  %.4g %s' % ('${names[$q]}', d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value'], d['unit']))"
  done
done
