#!/bin/bash
# tools/refresh_profiles.sh <round tag, e.g. r03> [all|pmc|stats] : on the GPU box, from the repo root.
# Writes small summaries under gpurun_out/profiles_<tag>/ (tools/install_profiles.py copies them into profiles/).
#   <tag>_bench_default.json       python bench.py (the driver's default invocation)
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats over bench.py --steps 20 --warmup 3 --no-cpu-baseline
#   <tag>_bench_headline_kernel_stats.csv + _headline_only.json   the same over bench.py ... --no-legs --score-tiles 0
#   <tag>_pmc_step_<leg>.json (+ .meta.json)   tools/pmc_groups.py (one rocprofv3 --pmc pass per counter group, --kernel-trace
#                                  only) over tools/step_prof.py at the shape of EVERY step leg of bench.py: the HBM traffic
#                                  bench.py's roofline objects quote (profiles/traffic.json)
#   <tag>_pmc_spmm.json            the same over tools/spmm_prof.py (LightGCN propagation product); _users / _items: its two halves
#   <tag>_pmc_scoring.json         over tools/score_prof.py for the scoring kernels
set -u
tag=${1:-r04}
what=${2:-all}
root=$(pwd)
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$what" = "all" ] || [ "$what" = "stats" ]; then
python3 bench.py > "$out/${tag}_bench_default.json" 2> "$out/bench.err"
cd /tmp
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 "$root/tools/prof_summarize.py" /tmp/prof_stats "$out/${tag}_bench" > /dev/null
# the headline alone (no legs, no scoring): here the step kernel's average IS the headline's
rm -rf /tmp/prof_head
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-legs --score-tiles 0 > "$out/${tag}_bench_headline_only.json" 2> /dev/null
python3 "$root/tools/prof_summarize.py" /tmp/prof_head "$out/${tag}_bench_headline" > /dev/null
cd "$root"
fi
if [ "$what" = "all" ] || [ "$what" = "pmc" ]; then
# full counter set on the three legs the design discussion leans on, the traffic counters on every other leg
G="FETCH_SIZE;WRITE_SIZE;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
T="FETCH_SIZE;WRITE_SIZE;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"
leg() {   # leg <name> <groups> <env assignments or -> <step_prof args...>
  local name=$1 groups=$2 envs=$3; shift 3
  ( [ "$envs" != "-" ] && export $envs; export STEP_PROF_META="$out/${tag}_pmc_step_${name}.meta.json"
    PMC_GROUPS="$groups" python3 tools/pmc_groups.py "$out/${tag}_pmc_step_${name}.json" bpr_step,apply_item,bucket_,bpr_sample -- python3 tools/step_prof.py "$@" > /dev/null 2>&1 )
}
leg B1M_blocked   "$G" - 1000000 8 12
leg B65536_plain  "$G" - 65536 0 60
leg B1M_iid       "$G" - 1000000 0 12
leg B1M_uniform   "$T" "POP=uniform" 1000000 8 12
leg B1M_d64       "$T" "DIM=64" 1000000 8 12
leg config3_slice "$T" "USERS=1250000 ITEMS=1000000 DEG=10" 1250000 8 8
leg B4096_plain   "$T" - 4096 0 100
leg B16384_plain  "$T" - 16384 0 100
leg B262144       "$T" - 262144 8 30
leg B1M_ranges3   "$T" "CHUNKS=3" 1000000 8 12
leg B1M_ranges2   "$T" "CHUNKS=2" 1000000 8 12
leg config3_ranges2 "$T" "USERS=1250000 ITEMS=1000000 DEG=10 CHUNKS=2" 1250000 8 8
( export STEP_PROF_META="$out/${tag}_pmc_spmm.meta.json"
  PMC_GROUPS="$T;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" python3 tools/pmc_groups.py "$out/${tag}_pmc_spmm.json" spmm_csr,spmm_hot,hot_rows_edge,zero_split -- python3 tools/spmm_prof.py > /dev/null 2>&1 )
for half in users items; do
( export SPMM_HALF=$half
  PMC_GROUPS="$T;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" python3 tools/pmc_groups.py "$out/${tag}_pmc_spmm_${half}.json" spmm_csr,spmm_hot,hot_rows_edge,zero_split -- python3 tools/spmm_prof.py > /dev/null 2>&1 )
done
S="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;FETCH_SIZE;WRITE_SIZE"
PMC_GROUPS="$S" LANES=2 python3 tools/pmc_groups.py "$out/${tag}_pmc_scoring.json" score_,merge_cand,topk_rows,sample_tau,permute_items -- python3 tools/score_prof.py 16 > /dev/null 2>&1
fi
ls -la "$out"
