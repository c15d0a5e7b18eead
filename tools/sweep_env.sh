#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... : run the default bench once per value of an environment knob
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 40 --warmup 4 --no-cpu-baseline --score-tiles 0 2>/dev/null > /tmp/b.json
  python - "$var=$v" <<'PY'
import sys, json
d = json.loads(open('/tmp/b.json').readline())
print(sys.argv[1], round(d["value"] / 1e9, 3), "Gtriplets/s", round(d["ms_per_step"] * 1e3, 1), "us/step, kernel", round(d["roofline"].get("kernel_ms", 0) * 1e3, 1), "us")
PY
done
