#!/bin/bash
# The random-shapes parity tests with OTHER seeds and more trials than the committed suite runs (tests/conftest.py: fuzz()).
#   tools/fuzz_campaign.sh <first seed> <last seed> <trial multiplier>     -> gpurun_out/fuzz_campaign.txt
# Every failing problem prints its own context line (shape, flags, seed) -- it goes into tests/ as a fixed case once understood.
mkdir -p gpurun_out
out=gpurun_out/fuzz_campaign.txt
: > $out
for seed in $(seq ${1:-1} ${2:-4}); do
    echo "== RSX_FUZZ_SEED=$seed RSX_FUZZ_TRIALS=${3:-4}" >> $out
    RSX_FUZZ_SEED=$seed RSX_FUZZ_TRIALS=${3:-4} timeout 1500 python -m pytest -q -m gpu -p no:cacheprovider --no-header --tb=short -s \
        "tests/test_gpu_parity.py::test_step_kernels_on_random_shapes" "tests/test_gpu_parity.py::test_fused_score_topk_on_random_shapes" \
        "tests/test_gpu_chunks.py::test_chunked_sampler_on_random_shapes" "tests/test_gpu_model.py::test_item_cdf_buckets_on_random_shapes" \
        "tests/test_gpu_parity.py::test_score_mask_topk_on_random_shapes" "tests/test_lightgcn.py::test_hip_spmm_on_random_graphs" \
        "tests/test_gpu_chunks.py::test_native_loop_on_random_shapes" "tests/test_gpu_parity.py::test_dense_gradient_paths_on_random_shapes" \
        "tests/test_gpu_parity.py::test_deterministic_step_on_random_shapes" "tests/test_gpu_model.py::test_model_on_random_shapes" \
        "tests/test_lightgcn.py::test_hip_lightgcn_on_random_graphs" \
        "tests/test_gpu_csc.py::test_csc_blob_is_the_transposed_matrix" "tests/test_gpu_csc.py::test_csc_sampler_on_random_shapes" \
        "tests/test_gpu_csc.py::test_csc_sampler_with_item_ranges_on_random_shapes" \
        2>&1 | grep -v "amdgpu.ids" | tail -40 >> $out
done
grep -c passed $out
