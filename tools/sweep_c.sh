# tools/sweep_c.sh : the headline (two rounds) and the d = 64 leg against the item block c of the stratified negatives (RSX_NEG_BLOCK_EXACT)
for round in 1 2; do
for c in 2 3 4 6; do
  RSX_NEG_BLOCK_EXACT=$c python3 bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('c=$c round $round  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))"
done; done
for c in 2 3 4; do
  RSX_NEG_BLOCK_EXACT=$c python3 bench.py --dim 64 --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('d=64 c=$c  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))"
done
