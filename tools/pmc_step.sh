#!/bin/bash
# tools/pmc_step.sh : instruction mix / issue counters of the step kernel (bench workload, one pass per counter group)
root=$(pwd); export TMPDIR=/tmp; cd /tmp
args="--steps 10 --warmup 2 --no-cpu-baseline --score-tiles 0 --small-batch 0"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVES SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  RSX_INLINE_SAMPLER=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc$i -- python3 $root/bench.py $args > /dev/null 2>&1
  python3 $root/tools/prof_summarize.py /tmp/pmc$i /tmp/pmc_out$i > /dev/null 2>&1
  python3 - /tmp/pmc_out${i}_counters.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no counters", e); sys.exit(0)
for k, v in d.items():
    if "bpr_step_blocked" in k:
        print({a: round(b) for a, b in v.items()})
PY
done
