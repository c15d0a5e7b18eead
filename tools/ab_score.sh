#!/bin/bash
# tools/ab_score.sh [-r rounds] lib-or-variant ... : same-box A/B of librsx builds on the scoring leg (tools/score_bench.py: 64 x 1024
# users x 100K items, mask + top-50, and the dense 1024 x I product).  `tree` = the library in the tree; any other name = a variant
# built by tools/build_variant.py (recsys_pytorch_amd/build/variants/librsx_<name>.so).
rounds=2
while getopts "r:" o; do case $o in r) rounds=$OPTARG;; esac; done
shift $((OPTIND - 1))
root=$(cd "$(dirname "$0")/.." && pwd)
for round in $(seq 1 $rounds); do
  for name in "$@"; do
    lib=$root/recsys_pytorch_amd/build/variants/librsx_$name.so; [ "$name" = "tree" ] && lib=$root/recsys_pytorch_amd/librsx.so
    RSX_LIB=$lib timeout 300 python3 "$root/tools/score_bench.py" 2>/dev/null | tail -1
  done
done
