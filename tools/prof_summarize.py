"""tools/prof_summarize.py -- condense rocprofv3 CSV output into small summaries.

    python tools/prof_summarize.py <rocprof_dir> <out_prefix>

Writes <out_prefix>_kernel_stats.csv (copy of the --stats table, if present) and
<out_prefix>_counters.json (per kernel name: launches, mean of each PMC counter
summed over dimensions/XCDs per dispatch).  The raw traces are deleted afterwards."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def main(src, out):
    files = glob.glob(os.path.join(src, "**", "*.csv"), recursive=True)
    for f in files:
        if f.endswith("kernel_stats.csv"):
            shutil.copy(f, out + "_kernel_stats.csv")
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> dispatch -> counter -> sum
    for f in files:
        if not f.endswith("counter_collection.csv"):
            continue
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                disp = row.get("Dispatch_Id", "0")
                per[k][disp][row.get("Counter_Name", "?")] += float(row.get("Counter_Value", 0) or 0)
    summ = {}
    for k, disps in per.items():
        acc = defaultdict(list)
        for d, cs in disps.items():
            for c, v in cs.items():
                acc[c].append(v)
        summ[k] = {"launches": len(disps), **{c: sum(v) / len(v) for c, v in acc.items()}}
    if summ:
        json.dump(summ, open(out + "_counters.json", "w"), indent=1, sort_keys=True)
    # durations from the kernel trace, if present
    dur = defaultdict(list)
    for f in files:
        if f.endswith("kernel_trace.csv"):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    try:
                        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                    except (KeyError, ValueError):
                        pass
    if dur:
        json.dump({k: {"launches": len(v), "mean_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)}
                   for k, v in dur.items()}, open(out + "_durations.json", "w"), indent=1, sort_keys=True)
    shutil.rmtree(src, ignore_errors=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
