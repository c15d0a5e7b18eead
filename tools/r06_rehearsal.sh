# the driver's round-end sequence on one fresh box: the -m gpu suite (as the driver runs it), smoke(), the default bench line
O=$(pwd)/gpurun_out/r06; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/rehearsal_suite.log 2>&1; echo "suite rc=$?"
tail -4 $O/rehearsal_suite.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/rehearsal_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/rehearsal_smoke.log
( time python bench.py ) > $O/rehearsal_bench.log 2> $O/rehearsal_bench.err; echo "bench rc=$?"
tail -1 $O/rehearsal_bench.log > $O/rehearsal_bench.json
python - <<'P'
import json
d=json.load(open("gpurun_out/r06/rehearsal_bench.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("achieved_over_copy_rate"), r.get("hbm_utilisation_e2e"), r.get("hbm_utilisation_e2e_over_copy_rate"), r.get("traffic_stale"))
P
tail -5 $O/rehearsal_bench.err
