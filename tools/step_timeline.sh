#!/bin/bash
# tools/step_timeline.sh <B> <neg_block> : start/end of every kernel of a few steps of the native loop (rocprofv3 --kernel-trace)
root=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/stl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -- python3 $root/tools/step_prof.py $1 $2 30 > /dev/null 2>&1
python3 $root/tools/timeline.py /tmp/stl
