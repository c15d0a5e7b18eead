# round 6, run 7: the longest rows of the LightGCN product by scatter (rsx_spmm_hot_rows): parity tests, then the product / the step with and without it
O=$(pwd)/gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_lightgcn.py tests/test_gpu_model.py -m gpu -x -q -k "lightgcn or spmm or small_batches" > $O/run7_tests.log 2>&1; echo "tests rc=$?" | tee $O/run7_rc.txt
tail -4 $O/run7_tests.log
rm -f $O/spmm_hot.txt
for hot in 0 1 0 1; do
  echo "RSX_SPMM_HOT=$hot" >> $O/spmm_hot.txt
  RSX_SPMM_HOT=$hot timeout 300 python tools/bench_lightgcn.py 2>&1 | grep -v amdgpu.ids >> $O/spmm_hot.txt
done
cat $O/spmm_hot.txt
