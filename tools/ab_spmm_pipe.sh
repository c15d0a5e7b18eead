#!/bin/bash
# tools/ab_spmm_pipe.sh : LightGCN product with the next group's (value, index) pairs requested ahead (RSX_SPMM_PIPELINE=1) vs not
for round in 1 2 3; do for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_spmmpipe.so; do echo -n "$(basename $l): "; RSX_LIB=$(pwd)/$l timeout 600 python3 tools/bench_lightgcn.py 2>/dev/null | grep -E "spmm:|train_step" | tr '\n' ' '; echo; done; done
