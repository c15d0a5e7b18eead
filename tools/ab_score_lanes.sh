#!/bin/bash
# tools/ab_score_lanes.sh : passes of the fused scoring path in flight (rsx_set_option score_lanes), two rounds
for round in 1 2; do for l in 1 2 3 4; do echo -n "lanes $l: "; RSX_SCORE_LANES=$l timeout 300 python3 tools/score_bench.py 2>/dev/null | tail -1; done; done
