# round 6, run 5: the CSC tests on the form without look-back; the row-marked apply of small batches (tests + what it buys per batch size)
O=$(pwd)/gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_csc.py tests/test_gpu_model.py -m gpu -x -q -k "csc or marked_rows or native_trainer or small_batches" > $O/run5_tests.log 2>&1; echo "tests rc=$?" | tee $O/run5_rc.txt
tail -5 $O/run5_tests.log
rm -f $O/touched_apply.txt
for b in 256 4096 16384 65536; do
  for opt in 0 1 2 0 2; do
    RSX_TOUCHED_APPLY=$opt timeout 300 python bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 200 --warmup 10 --batch $b --neg-block 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('B=$b touched_apply=$opt  %8.2f us/step  kernel %7.2f us  %.4g triplets/s' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))" | tee -a $O/touched_apply.txt
  done
done
