# round 6, run 11: blocks of rsx_mesh_alloc never die: the stress with every mesh's tables dropped and torch's cache emptied between meshes, then the
# whole -m gpu suite on the final tree
O=$(pwd)/gpurun_out/r06; mkdir -p $O
rm -f $O/mesh_stress3.jsonl
for cfg in "--world 4 --loops 10 --empty-cache --mesh-memory" "--world 8 --loops 3 --empty-cache --mesh-memory" "--world 3 --loops 10 --empty-cache"; do
  timeout 500 python tools/mesh_stress.py $cfg --out $O/mesh_stress3.jsonl > $O/mesh_stress3_last.log 2>&1; echo "rc=$? stress $cfg" | tee -a $O/run11_rc.txt
done
python3 - <<'PY'
import json
for line in open('gpurun_out/r06/mesh_stress3.jsonl'):
    d = json.loads(line)
    print({k: d[k] for k in ('world', 'loops', 'mesh_memory', 'empty_cache', 'status', 'failures', 'retries')}, [ (r, v['meshes'], v['seconds'], [f[3:5] for f in v['fails'][:2]]) for r, v in sorted(d['ranks'].items())][:2])
PY
( time RSX_SAVE_8RANK_LINE=$O/bench_8ranks_one_gpu.json timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $O/suite_final.log 2>&1; echo "suite rc=$?" | tee -a $O/run11_rc.txt
tail -6 $O/suite_final.log
