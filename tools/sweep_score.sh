#!/bin/bash
# tools/sweep_score.sh VAR v1 v2 ... : scoring leg of the bench once per value of an environment knob
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --small-batch 0 --score-tiles ${TILES:-64} 2>/dev/null > /tmp/b.json
  python - "$var=$v" <<'PY'
import sys, json
d = json.loads(open('/tmp/b.json').readline())["scoring"]
print(sys.argv[1], round(d["value"] / 1e11, 3), "e11 scores/s", round(d["roofline"]["frac"], 3), "of MFMA peak")
PY
done
