#!/bin/bash
# tools/score_trace.sh : per-kernel durations of one fused scoring call (rocprofv3 --kernel-trace --stats), full and
# with the dev library's "hits detected, nothing stored" switch; LANES = passes in flight
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for abl in 0 2; do
  for lanes in 1 2; do
    rm -rf /tmp/sct
    ABL=$abl LANES=$lanes RSX_LIB=$root/recsys_pytorch_amd/librsx_dev.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sct -- python3 $root/tools/score_prof.py 16 > /dev/null 2>&1
    echo "== ablate=$abl lanes=$lanes"
    python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/sct/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(s in n for s in ("score_tile", "merge_cand", "topk_rows", "fill", "Memset", "memset")):
            print(f'   {n.split("(")[0][-48:]:48s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e3:10.1f} us')
PY
  done
done
