#!/bin/bash
# tools/ab_libs.sh lib1 lib2 ... : same-box A/B of librsx builds (stand-alone step kernel + default bench)
keep=/tmp/librsx_keep.so
cp recsys_pytorch_amd/librsx.so $keep
for round in 1 2; do
for l in "$@"; do
  cp $l recsys_pytorch_amd/librsx.so
  echo "== $l"
  timeout 120 python tools/two_pass_times.py 2>/dev/null | head -1
  timeout 120 bash tools/sweep_env.sh RSX_X 0
done
done
cp $keep recsys_pytorch_amd/librsx.so
