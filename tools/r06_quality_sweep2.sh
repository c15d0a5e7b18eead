O=gpurun_out/r06; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/sampler_quality.py "$@" > $O/sq_$name.txt 2>> $O/sq.err; echo "== $name: $@"; grep "^#" $O/sq_$name.txt | grep -B0 -A3 "NDCG@10:" ; }
for c in 2 4 8 16 32; do run c$c --arms blocked --seeds 8 --epochs 1000 --every 250 --force-block $c; done
run iid1000 --arms iid --seeds 8 --epochs 1000 --every 250
