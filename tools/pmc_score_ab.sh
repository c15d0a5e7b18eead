#!/bin/bash
# tools/pmc_score_ab.sh lib1 lib2 ... : MFMA-busy / wait counters of the scoring kernels for several librsx builds (one rocprofv3 --pmc pass per group)
S="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
for l in "$@"; do
  n=$(basename $l .so)
  RSX_LIB=$(pwd)/$l PMC_GROUPS="$S" LANES=2 python3 tools/pmc_groups.py gpurun_out/pmc_score_$n.json score_tile -- python3 tools/score_prof.py 16 > /dev/null 2>&1
  python3 - gpurun_out/pmc_score_$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if 'true' in k:
        print(sys.argv[2], k, 'us', v.get('mean_us'), 'mfma_busy %.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES']/(4*v['SQ_BUSY_CU_CYCLES'])), 'waves/SIMD %.2f' % (v['SQ_WAVE_CYCLES']*4/(4*v['SQ_BUSY_CU_CYCLES'])), 'wait_inst %.2f' % (v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']), 'VALU', v.get('SQ_INSTS_VALU'), 'LDS', v.get('SQ_INSTS_LDS'), 'bank_conflict', v.get('SQ_LDS_BANK_CONFLICT'))
PY
done
