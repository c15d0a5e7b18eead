#!/bin/bash
# tools/exp_chunks.sh : where does the chunked step's time go?  timeline of the pipelined and of the serial form (same box)
echo "== unchunked"; bash tools/step_timeline.sh 1000000 8
echo "== chunks 4, pipelined"; CHUNKS=4 bash tools/step_timeline.sh 1000000 8
echo "== chunks 4, serial (RSX_CHUNK_SERIAL: waits + applies after the kernel, run stream)"; RSX_CHUNK_SERIAL=1 CHUNKS=4 bash tools/step_timeline.sh 1000000 8
echo "== A/B"; for c in 0 4; do for ser in "" 1; do
  [ "$c" = 0 ] && [ "$ser" = 1 ] && continue
  ( [ -n "$ser" ] && export RSX_CHUNK_SERIAL=1; python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --chunks $c --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('chunks', d['config']['item_chunks'], 'serial' if os.environ.get('RSX_CHUNK_SERIAL') else 'pipelined', 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms'])" )
done; done
