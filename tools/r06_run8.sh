O=$(pwd)/gpurun_out/r06; mkdir -p $O; root=$(pwd); export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_hot
RSX_SPMM_HOT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hot -- python3 $root/tools/bench_lightgcn.py > $O/hot_trace_line.txt 2>&1
python3 $root/tools/prof_summarize.py /tmp/prof_hot $O/hot_trace > /dev/null 2>&1
cd $root
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r06/hot_trace_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('spmm', 'hot_rows', 'zero_split', 'adam', 'scale', 'bpr_grad', 'mark_batch')):
        print(r['Name'][:90], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
