#!/bin/bash
# tools/ab_step19.sh : grid cap of the LightGCN product (workgroups per CU); rows per pass of the fused scoring path
for round in 1 2; do
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_spg64.so recsys_pytorch_amd/build/variants/librsx_spg256.so recsys_pytorch_amd/build/variants/librsx_spg1024.so; do echo -n "$(basename $l): "; RSX_LIB=$(pwd)/$l timeout 600 python3 tools/bench_lightgcn.py 2>/dev/null | grep -E "spmm:|train_step" | tr '\n' ' '; echo; done
for l in recsys_pytorch_amd/librsx.so recsys_pytorch_amd/build/variants/librsx_fr4096.so recsys_pytorch_amd/build/variants/librsx_fr16384.so; do RSX_LIB=$(pwd)/$l timeout 300 python3 tools/score_bench.py 2>/dev/null | tail -1; done
done
