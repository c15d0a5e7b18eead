#!/bin/bash
# tools/pmc_score.sh [tiles] : PMC counter groups of the scoring kernels (one rocprofv3 pass per group;
# counters only, with --kernel-trace), condensed per kernel into gpurun_out/pmc_score_<cfg>.json
root=$(pwd); export TMPDIR=/tmp; mkdir -p $root/gpurun_out; cd /tmp
for cfg in "${@:-16}"; do
  tag=$(echo $cfg | tr ' ' '_')
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs_${tag}_$i -- python3 $root/tools/score_prof.py $cfg > /dev/null 2>&1
    python3 $root/tools/prof_summarize.py /tmp/pmcs_${tag}_$i /tmp/pmcs_out_${tag}_$i > /dev/null 2>&1
  done
  python3 - $tag $root <<'PY'
import json, sys, glob
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob(f"/tmp/pmcs_out_{tag}_*_counters.json")):
    for k, v in json.load(open(f)).items():
        if any(s in k for s in ("score_", "merge_cand", "topk_rows")):
            out.setdefault(k.split("(")[0][-60:], {}).update({a: round(b) for a, b in v.items()})
for f in sorted(glob.glob(f"/tmp/pmcs_out_{tag}_1_durations.json")):
    for k, v in json.load(open(f)).items():
        if any(s in k for s in ("score_", "merge_cand", "topk_rows")):
            out.setdefault(k.split("(")[0][-60:], {})["mean_us"] = round(v["mean_ns"] / 1e3, 1)
json.dump(out, open(f"{sys.argv[2]}/gpurun_out/pmc_score_{tag}.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(tag, k, v)
PY
done
