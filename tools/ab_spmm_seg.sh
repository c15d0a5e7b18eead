#!/bin/bash
# tools/ab_spmm_seg.sh : longest row segment of the LightGCN product (rows above it are split and combined with atomics), plan order vs longest first
for round in 1 2; do for so in 0 1; do for ms in 1024 2048 4096 16384; do echo -n "sort $so max_seg $ms: "; RSX_SPMM_SORT=$so RSX_SPMM_MAX_SEG=$ms timeout 600 python3 tools/bench_lightgcn.py 2>/dev/null | grep -E "spmm:|train_step|segments" | tr '\n' ' ' | sed 's/graph build + upload [0-9.]*s, nnz(A)=40000000, //; s/algorithmic 21.93 GB -> //'; echo; done; done; done
