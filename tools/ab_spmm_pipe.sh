#!/bin/bash
# tools/ab_spmm_pipe.sh lib... : LightGCN product between librsx builds, three rounds
for round in 1 2 3; do for l in "$@"; do echo -n "$(basename $l): "; RSX_LIB=$(pwd)/$l timeout 600 python3 tools/bench_lightgcn.py 2>/dev/null | grep -E "spmm:|train_step" | tr '\n' ' '; echo; done; done
