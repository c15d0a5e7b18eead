mkdir -p gpurun_out/r06
O=gpurun_out/r06/mesh_stress.jsonl
rm -f $O
for cfg in "--world 4 --loops 20 --no-post-barrier" "--world 4 --loops 20" "--world 8 --loops 20" "--world 3 --loops 20" "--world 2 --loops 20" "--world 8 --loops 10 --own-memory" "--world 8 --loops 20 --no-post-barrier"; do
  timeout 600 python tools/mesh_stress.py $cfg --out $O > gpurun_out/r06/mesh_stress_last.log 2>&1
  echo "rc=$? $cfg" >> gpurun_out/r06/mesh_stress_rc.txt
done
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > gpurun_out/r06/suite_durations.log 2>&1
echo "suite rc=$?" >> gpurun_out/r06/mesh_stress_rc.txt
tail -5 gpurun_out/r06/suite_durations.log
cat gpurun_out/r06/mesh_stress_rc.txt
