#!/bin/bash
# tools/refresh_profiles.sh <round tag, e.g. r02> : on the GPU box, from the repo root.
# Writes small summaries under gpurun_out/profiles_<tag>/ (tools/install_profiles.py copies them into profiles/).
#   <tag>_bench_default.json       python bench.py (the driver's default invocation)
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats over bench.py --steps 20 --warmup 3 --no-cpu-baseline
#   <tag>_bench_headline_kernel_stats.csv + _headline_only.json   the same over bench.py ... --no-legs --score-tiles 0
#   <tag>_pmc_<leg>.json           tools/pmc_groups.py (one rocprofv3 --pmc pass per counter group, --kernel-trace only)
#                                  over tools/step_prof.py at the headline, base-batch and iid-negative shapes,
#                                  and over tools/score_prof.py for the scoring kernels
set -u
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ "${2:-all}" != "pmc" ]; then
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
G="FETCH_SIZE;WRITE_SIZE;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
PMC_GROUPS="$G" python3 tools/pmc_groups.py "$out/${tag}_pmc_step_B1M_blocked.json" bpr_step,apply_item,bucket_ -- python3 tools/step_prof.py 1000000 8 12 > /dev/null 2>&1
PMC_GROUPS="$G" python3 tools/pmc_groups.py "$out/${tag}_pmc_step_B65536_plain.json" bpr_step,apply_item,bpr_sample -- python3 tools/step_prof.py 65536 0 60 > /dev/null 2>&1
PMC_GROUPS="$G" python3 tools/pmc_groups.py "$out/${tag}_pmc_step_B1M_iid.json" bpr_step,apply_item,bucket_ -- python3 tools/step_prof.py 1000000 0 12 > /dev/null 2>&1
S="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32;FETCH_SIZE;WRITE_SIZE"
PMC_GROUPS="$S" LANES=2 python3 tools/pmc_groups.py "$out/${tag}_pmc_scoring.json" score_,merge_cand,topk_rows,mask_seen,take_tau,sample_tau -- python3 tools/score_prof.py 16 > /dev/null 2>&1
ls -la "$out"
