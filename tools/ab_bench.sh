#!/bin/bash
# tools/ab_bench.sh lib1 lib2 ... : same-box A/B of librsx builds on the default bench (3 rounds)
keep=/tmp/librsx_keep.so
cp recsys_pytorch_amd/librsx.so $keep
for round in 1 2 3; do
for l in "$@"; do
  cp $l recsys_pytorch_amd/librsx.so
  timeout 120 bash tools/sweep_env.sh "$(basename $l)" 0
done
done
cp $keep recsys_pytorch_amd/librsx.so
