# headline step time against the item block c of the stratified negatives (the speed side of tools/sampler_quality.py's quality side)
O=gpurun_out/r06; mkdir -p $O; : > $O/block_speed.txt
for c in 2 3 4 6 8 12 16; do
  RSX_NEG_BLOCK_EXACT=$c python bench.py --no-legs --no-lightgcn --no-cpu-baseline --score-tiles 1 --neg-block 16 2>/dev/null | tail -1 > $O/bs_$c.json
  python - $c <<'P' >> $O/block_speed.txt
import json, sys
d = json.load(open(f"gpurun_out/r06/bs_{sys.argv[1]}.json")); r = d["roofline"]
print(f"c={sys.argv[1]:>2s}  neg_block {d['config'].get('neg_block')}  {d['ms_per_step']*1e3:7.1f} us/step  kernel {r['kernel_ms']*1e3:7.1f} us  {d['value']:.4g} triplets/s")
P
done
cat $O/block_speed.txt
timeout 600 python tools/sampler_quality.py --arms blocked --seeds 4 --epochs 500 --every 250 --force-block 8 | grep "^#" | head -3
