O=$(pwd)/gpurun_out/r06; mkdir -p $O
( time RSX_SAVE_8RANK_LINE=$O/bench_8ranks_one_gpu.json timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $O/suite_final.log 2>&1; echo "suite rc=$?"
tail -6 $O/suite_final.log
