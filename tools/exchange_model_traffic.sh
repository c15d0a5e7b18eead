# tools/exchange_model_traffic.sh : the one-GPU schedule model (tools/exchange_model.sh) with the exchange stand-in also moving the message through HBM
# only the arms with the traffic stand-in (and their plain twins) at a few delays
export RSX_LIB=$(pwd)/recsys_pytorch_amd/librsx_dev.so RSX_FORCE_SHARDED=1 MASTER_PORT=29641
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for d in $DELAYS; do env "${envs[@]}" RSX_EXCHANGE_DELAY_US=$d python3 bench.py $A "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-44s delay %4d us  %7.1f us/step  (kernels %.1f)' % ('$label', $d, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"; done; }
A="--no-legs --score-tiles 0 --no-cpu-baseline --steps 40 --warmup 5"; DELAYS="100 200 300 400"
echo "# headline shape (51 MB exchange)"
run "one pass" RSX_TWO_PASS=0 --
run "one pass, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 --
run "item ranges x2" RSX_TWO_PASS=0 -- --chunks 2
run "item ranges x2, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 -- --chunks 2
A="--users 1250000 --items 1000000 --degree 10 --batch 1250000 --no-legs --score-tiles 0 --no-cpu-baseline --steps 12 --warmup 3"; DELAYS="500 1000"
echo "# configs[3] slice (512 MB exchange)"
run "one pass" RSX_TWO_PASS=0 --
run "one pass, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 --
run "item ranges x2" RSX_TWO_PASS=0 -- --chunks 2
run "item ranges x2, stand-in with traffic" RSX_TWO_PASS=0 RSX_EXCHANGE_TRAFFIC=1 -- --chunks 2
