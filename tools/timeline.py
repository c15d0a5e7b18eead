"""tools/timeline.py <rocprof_dir> : print a per-kernel timeline of a few bench steps (development aid)"""
import csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], "s%s/q%s" % (r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))))
rows.sort()
names = {"bpr_step_blocked": "STEP", "bpr_step_kernel": "STEP", "bpr_sample_kernel": "sample", "apply_item": "apply", "sample_ui16": "ui", "sample_neg16": "neg", "rocprim": "sort", "fold_hot": "fold", "bucket_chunk": "chunk", "bucket_sort": "bsort", "fill": "memset", "wait_progress": "wait", "fold_hot_range": "foldr"}
steps = [i for i, r in enumerate(rows) if "bpr_step_blocked" in r[2] or "bpr_step_kernel" in r[2]]
if len(steps) > 14:
    import os
    lo, hi = steps[int(os.environ.get("TL_FROM", 10))], steps[int(os.environ.get("TL_TO", 13))]
    t0 = rows[lo][0]
    for s, e, n, q in rows[lo - 6:hi + 1]:
        tag = next((v for k, v in names.items() if k in n), None)
        if tag:
            print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {q}  {tag}")
