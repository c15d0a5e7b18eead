# round 6, run 10: the mesh over memory from rsx_mesh_alloc -- the tests three times over, the stress with and without torch.cuda.empty_cache()
# between meshes (does a re-allocated pooled segment provoke the refusal?), on borrowed and on library memory
O=$(pwd)/gpurun_out/r06; mkdir -p $O
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_sharded_gloo.py tests/test_bench_contract.py -m gpu -q -x -k "mesh or direct or plain_command" > $O/run10_tests_$i.log 2>&1; echo "tests pass $i rc=$?" | tee -a $O/run10_rc.txt
  tail -2 $O/run10_tests_$i.log
done
rm -f $O/mesh_stress2.jsonl
for cfg in "--world 4 --loops 10 --empty-cache" "--world 3 --loops 10 --empty-cache" "--world 4 --loops 10 --empty-cache --mesh-memory" "--world 8 --loops 3 --mesh-memory" "--world 4 --loops 20 --mesh-memory"; do
  timeout 400 python tools/mesh_stress.py $cfg --out $O/mesh_stress2.jsonl > $O/mesh_stress2_last.log 2>&1; echo "rc=$? stress $cfg" | tee -a $O/run10_rc.txt
done
python3 - <<'PY'
import json
for line in open('gpurun_out/r06/mesh_stress2.jsonl'):
    d = json.loads(line)
    print({k: d[k] for k in ('world', 'loops', 'mesh_memory', 'empty_cache', 'status', 'failures', 'retries')}, [ (r, v['meshes'], v['seconds'], [f[3:5] for f in v['fails'][:2]]) for r, v in sorted(d['ranks'].items())][:3])
PY
