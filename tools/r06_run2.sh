# round 6, run 2: FIRST the whole -m gpu suite on the fresh box (cold: what the driver's run sees), then the CSC-walk sampler
mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests -m gpu -q --durations=40 ) > gpurun_out/r06/suite_cold.log 2>&1; echo "cold suite rc=$?" > gpurun_out/r06/run2_rc.txt
tail -4 gpurun_out/r06/suite_cold.log
for cfg in "--world 4 --loops 5" "--world 4 --loops 5 --no-post-barrier" "--world 8 --loops 2"; do
  timeout 300 python tools/mesh_stress.py $cfg --out gpurun_out/r06/mesh_phases.jsonl > /dev/null 2>&1; echo "rc=$? stress $cfg" >> gpurun_out/r06/run2_rc.txt
done
O=gpurun_out/r06
for a in "" "--pop uniform" "--users 1250000 --items 1000000 --degree 10 --neg-block 0"; do
  timeout 300 python tools/sampler_csc_time.py $a >> $O/sampler_csc_time.jsonl 2>> $O/sampler_csc_time.err
done
cat $O/sampler_csc_time.jsonl
ab() {  # ab <label> <bench args>
  for round in 1 2 3; do
    for csc in 0 1; do
      RSX_CSC_SAMPLER=$csc timeout 300 python bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$1 csc=$csc round $round  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))" | tee -a $O/ab_csc.txt
    done
  done
}
ab d128 ""
ab d64 "--dim 64"
for v in r8 r2; do
  RSX_LIB=recsys_pytorch_amd/build/variants/librsx_$v.so timeout 300 python bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('d128 rounds-variant $v  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))" | tee -a $O/ab_csc.txt
  RSX_LIB=recsys_pytorch_amd/build/variants/librsx_$v.so timeout 300 python bench.py --dim 64 --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('d64 rounds-variant $v  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))" | tee -a $O/ab_csc.txt
done
