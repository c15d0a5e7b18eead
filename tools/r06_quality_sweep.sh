O=gpurun_out/r06; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/sampler_quality.py "$@" > $O/sq_$name.txt 2>> $O/sq.err; echo "== $name: $@"; grep "^#" $O/sq_$name.txt | grep -v Recall -A0 | head -8; }
run long   --arms iid,blocked --seeds 8 --epochs 1500 --every 250
run lr05   --arms iid,blocked --seeds 8 --epochs 1200 --every 200 --lr 0.05
run lr02   --arms iid,blocked --seeds 8 --epochs 1500 --every 250 --lr 0.02
run d128   --arms iid,blocked --seeds 8 --epochs 600 --every 100 --dim 128
run big    --arms iid,blocked --seeds 6 --epochs 600 --every 100 --users 400000 --items 20000
