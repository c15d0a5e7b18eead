# the quality comparison AT THE HEADLINE SHAPE (1M users x 100K items, d = 128, batch = users) with the clock beside it
O=gpurun_out/r06; mkdir -p $O
run() { name=$1; shift; timeout 1500 python tools/sampler_quality.py "$@" > $O/sq_$name.txt 2>> $O/sq.err; echo "== $name: $@"; grep "^#" $O/sq_$name.txt | grep -v "Recall\|^#   .*Recall" ; }
run headline_lr05 --arms iid,blocked --users 1000000 --items 100000 --dim 128 --seeds 3 --epochs 400 --every 50 --lr 0.05
