# A/B of the item block on ONE box, alternating: headline (d = 128), d = 64, two item ranges
O=gpurun_out/r06; mkdir -p $O; : > $O/block_speed2.txt
one() { tag=$1; c=$2; shift 2
  RSX_NEG_BLOCK_EXACT=$c python bench.py --no-legs --no-lightgcn --no-cpu-baseline --score-tiles 1 --neg-block 16 "$@" 2>/dev/null | tail -1 > $O/bs.json
  python - "$tag" $c <<'P' >> $O/block_speed2.txt
import json, sys
d = json.load(open("gpurun_out/r06/bs.json")); r = d["roofline"]
print(f"{sys.argv[1]:10s} c={sys.argv[2]:>2s}  {d['ms_per_step']*1e3:7.1f} us/step  kernel {r['kernel_ms']*1e3:7.1f} us  regions {[round(x*1e3,1) for x in d['timed_regions']['ms_per_step_each']]}")
P
}
for rep in 1 2 3; do for c in 2 3 4 5; do one headline $c; done; done
for rep in 1 2; do for c in 2 3 4; do one d64 $c --dim 64; done; done
for rep in 1 2; do for c in 3 4 5; do one ranges2 $c --chunks 2; done; done
cat $O/block_speed2.txt
