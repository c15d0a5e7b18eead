#!/bin/bash
# tools/step_trace.sh <B> <neg_block> [steps] : per-kernel durations of the native loop (rocprofv3 --kernel-trace --stats)
root=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/stt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stt -- python3 $root/tools/step_prof.py $1 $2 ${3:-100} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/stt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f'   {n[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]:>6s}%')
PY
