# round 6, run 4: the CSC walk without a look-back (select into per-tile slices -> scan -> place -> negatives): tests, time alone (+ kernel trace),
# the step with it against the step with the bucket passes; what the per-launch timing events cost; the exchange schedules re-modelled
O=$(pwd)/gpurun_out/r06; mkdir -p $O; root=$(pwd)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_csc.py -x -q > $O/csc_tests2.log 2>&1; echo "csc tests rc=$?" | tee $O/run4_rc.txt
tail -3 $O/csc_tests2.log
rm -f $O/sampler_csc_time2.jsonl
for a in "" "--pop uniform" "--users 1250000 --items 1000000 --degree 10 --neg-block 0"; do
  timeout 300 python tools/sampler_csc_time.py $a >> $O/sampler_csc_time2.jsonl 2>> $O/sampler_csc_time2.err
done
cat $O/sampler_csc_time2.jsonl
cd /tmp; rm -rf /tmp/prof_csc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_csc -- python3 $root/tools/sampler_csc_time.py --iters 20 > /dev/null 2>&1
python3 $root/tools/prof_summarize.py /tmp/prof_csc $O/csc_trace2 > /dev/null 2>&1
cd $root
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r06/csc_trace2_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('csc_', 'bucket_')):
        print(r['Name'][:60], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
rm -f $O/ab_csc2.txt
ab() {  # ab <label> <bench args>
  for round in 1 2 3; do
    for csc in 0 1; do
      RSX_CSC_SAMPLER=$csc timeout 300 python bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$1 csc=$csc round $round  %8.1f us/step  kernel %8.1f us  %.4g' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, d['value']))" | tee -a $O/ab_csc2.txt
    done
  done
}
ab d128 ""
ab d64 "--dim 64"
ab config3 "--users 1250000 --items 1000000 --degree 10 --batch 1250000"
for te in 1 0 4 1 0 4; do
  RSX_CSC_SAMPLER=0 RSX_TIME_EVERY=$te timeout 300 python bench.py --no-legs --score-tiles 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('time_every=$te  %8.1f us/step  kernel %8.1f us (%d timed)' % (d['ms_per_step']*1e3, (r['kernel_ms'] or 0)*1e3, r['kernel_launches_timed']))" | tee -a $O/ab_csc2.txt
done
RSX_CSC_SAMPLER=0 timeout 1500 bash tools/exchange_model_schedules.sh > $O/exchange_model_schedules2.txt 2> $O/exchange_model_schedules2.err
cat $O/exchange_model_schedules2.txt
