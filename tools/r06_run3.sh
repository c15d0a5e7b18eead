# round 6, run 3: where the CSC walk's time goes (kernel trace of the stand-alone sampler timing), then the exchange schedules' HBM side at world 8
O=$(pwd)/gpurun_out/r06; mkdir -p $O; root=$(pwd)
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_csc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_csc -- python3 $root/tools/sampler_csc_time.py --iters 20 > $O/csc_trace_line.json 2> /dev/null
python3 $root/tools/prof_summarize.py /tmp/prof_csc $O/csc_trace > /dev/null 2>&1
ls /tmp/prof_csc/*/ 2>/dev/null | head
cd $root
head -12 $O/csc_trace_kernel_stats.csv 2>/dev/null | cut -c1-220
timeout 1500 bash tools/exchange_model_schedules.sh > $O/exchange_model_schedules.txt 2> $O/exchange_model_schedules.err
cat $O/exchange_model_schedules.txt
