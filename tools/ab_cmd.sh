#!/bin/bash
# tools/ab_cmd.sh "<command>" lib1 lib2 ... : run a command under each librsx build (2 rounds, same box)
cmd=$1; shift
keep=/tmp/librsx_keep.so
cp recsys_pytorch_amd/librsx.so $keep
for round in 1 2; do
for l in "$@"; do
  cp $l recsys_pytorch_amd/librsx.so
  echo "== $l"; timeout 200 bash -c "$cmd" 2>/dev/null
done
done
cp $keep recsys_pytorch_amd/librsx.so
