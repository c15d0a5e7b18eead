O=gpurun_out/r06; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/sampler_quality.py "$@" > $O/sq_$name.txt 2>> $O/sq.err; echo "== $name: $@"; grep "^#" $O/sq_$name.txt | grep -A2 "NDCG@10:" ; }
for c in 3 6; do run c$c --arms blocked --seeds 8 --epochs 1000 --every 250 --force-block $c; done
run lr02_16 --arms iid,blocked --seeds 16 --epochs 1500 --every 500 --lr 0.02
