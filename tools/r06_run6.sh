# round 6, run 6: the whole -m gpu suite on this tree (fresh box: cold), then the round's profiles (tools/refresh_profiles.sh r06)
O=$(pwd)/gpurun_out/r06; mkdir -p $O
( time RSX_SAVE_8RANK_LINE=$O/bench_8ranks_one_gpu.json timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $O/suite_final.log 2>&1; echo "suite rc=$?" | tee $O/run6_rc.txt
tail -6 $O/suite_final.log
timeout 3000 bash tools/refresh_profiles.sh r06 all; echo "refresh rc=$?" | tee -a $O/run6_rc.txt
ls gpurun_out/profiles_r06 | head -80
