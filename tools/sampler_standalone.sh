#!/bin/bash
# tools/sampler_standalone.sh : stand-alone durations of the sampler's kernels (rocprofv3 with a counter serialises the kernels)
export TMPDIR=/tmp
one() { ( export $1; PMC_GROUPS="GRBM_GUI_ACTIVE" python3 tools/pmc_groups.py gpurun_out/sampler_sa.json bucket_,bpr_step_blocked -- python3 tools/step_prof.py 1000000 8 12 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    k,_,v=l.partition(' {')
    try: d=json.loads('{'+v)
    except Exception: continue
    print('$1', k[:40], d.get('mean_us'))" ) ; }
one "CHUNKS=0"
one "CHUNKS=0 NEG_EXACT=3"
one "CHUNKS=3"
one "CHUNKS=2"
one "CHUNKS=4"
one "CHUNKS=3 NEG_EXACT=2"
one "CHUNKS=3 NEG_EXACT=4"
