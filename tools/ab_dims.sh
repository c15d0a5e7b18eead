#!/bin/bash
# tools/ab_dims.sh lib1 lib2 ... : same-box A/B of librsx builds on the default bench at d = 128, 64, 32
keep=/tmp/librsx_keep.so
cp recsys_pytorch_amd/librsx.so $keep
for round in 1 2; do
for l in "$@"; do
  cp $l recsys_pytorch_amd/librsx.so
  for d in 128 64 32; do
    python bench.py --steps 30 --warmup 4 --no-cpu-baseline --score-tiles 0 --small-batch 0 --dim $d 2>/dev/null > /tmp/b.json
    python - "$l d=$d" <<'PY'
import sys, json
d = json.loads(open('/tmp/b.json').readline())
print(sys.argv[1], round(d["value"] / 1e9, 3), "Gtriplets/s", round(d["ms_per_step"] * 1e3, 1), "us/step, kernel", round(d["roofline"]["kernel_ms"] * 1e3, 1))
PY
  done
done
done
cp $keep recsys_pytorch_amd/librsx.so
