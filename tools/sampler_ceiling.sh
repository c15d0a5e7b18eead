for dim in 128 64; do for rep in 0 1; do
RSX_LIB=$PWD/recsys_pytorch_amd/librsx_dev.so RSX_SAMPLER_REPLAY=$rep python bench.py --dim $dim --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('d=$dim replay=$rep  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))"
done; done
