#!/bin/bash
# tools/ab_score.sh lib1 lib2 ... : same-box A/B of the scoring leg between librsx builds (RSX_LIB), two rounds
for round in 1 2; do
for l in "$@"; do
  RSX_LIB=$l timeout 300 python3 tools/score_bench.py 2>/dev/null | tail -1
done
done
