"""Do the parity tests catch a 1 % error?  (`-m gpu`)

The development build of the library (librsx_dev.so, -DRSX_ABLATE; `python -m recsys_pytorch_amd.build --dev`, also built by
__graft_entry__.build()) can plant a 1 % error in the step kernels:
    mask 128   the user-row update  P[u] += 1.01 * lr * c * (Q[i] - Q[j])
    mask 256   the item gradients   G[i] += 1.01 * g * P[u],  G[j] -= 1.01 * g * P[u]
(both in bpr_step_kernel and in bpr_step_blocked_kernel; the shipped librsx.so has no such switch).  Selected parity tests
-- full size and oracle replays, through the blocked kernel, its TILE = false form, the plain kernel and the native loop --
are re-run in a child process against that library: with the error planted every one of them must FAIL on its update
assertion; with the mask at zero the same library must pass.  A parity suite that cannot see 1 % is not a parity suite.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "recsys_pytorch_amd", "librsx_dev.so")

FULL_SIZE = ["tests/test_gpu_model.py::test_full_size_sampled_step_invariants[128]",
             "tests/test_gpu_model.py::test_full_size_step_through_the_native_loop[headline]",
             "tests/test_gpu_model.py::test_full_size_step_through_the_native_loop[iid]",
             "tests/test_gpu_model.py::test_base_batch_65536_on_the_headline_tables"]
ORACLE = ["tests/test_gpu_model.py::test_sorted_blocked_sampled_path_replays_through_oracle[128-40000-5000-8-True]",
          "tests/test_gpu_model.py::test_sorted_runs_layout_replays_through_oracle",
          "tests/test_gpu_model.py::test_native_trainer_steps_replay_through_the_oracle",
          "tests/test_gpu_model.py::test_blocked_kernel_is_exact_on_foreign_triplets",
          "tests/test_gpu_chunks.py::test_chunked_native_steps_follow_the_range_rule_and_replay_through_the_oracle[30000-5000-128-30000-12-4-32]",
          # (the item ranges WITHOUT blocks: the TILE = false walk launched per range, round 4)
          "tests/test_gpu_chunks.py::test_chunked_native_steps_follow_the_range_rule_and_replay_through_the_oracle[30000-40000-128-30000-12-2-32]",
          "tests/test_gpu_parity.py::test_step_kernels_on_random_shapes",
          "tests/test_gpu_parity.py::test_bpr_step_matches_reference_golden[g1c_sgd_biglr_400x250_d128_b512]"]


def run_child(tests, mask, var="RSX_ABLATION"):
    env = dict(os.environ, RSX_LIB=DEV_LIB, COLUMNS="2000", **{var: str(mask)})     # (wide: -rf lines are not truncated)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--no-header", "--tb=line", "-rf",
                        *tests], cwd=ROOT, env=env, capture_output=True, text=True, timeout=550)
    return r.returncode, r.stdout[-20000:] + r.stderr[-2000:]


def outcomes(text, tests):
    """test id -> the reason pytest's short summary (-rf) gives for its failure, or None if it passed"""
    got = {t: None for t in tests}
    for line in text.splitlines():
        if line.startswith("FAILED "):
            for t in tests:
                if line.startswith(f"FAILED {t} ") or line.rstrip() == f"FAILED {t}":
                    got[t] = line[len(f"FAILED {t}"):].strip(" -") or "failed"
    return got


@pytest.fixture(scope="module")
def dev_lib():
    if not os.path.exists(DEV_LIB):
        from recsys_pytorch_amd import build
        build.build(dev=True)
    return DEV_LIB


def test_dev_library_without_a_planted_error_passes(dev_lib):
    tests = [FULL_SIZE[3], ORACLE[0], ORACLE[3]]
    rc, text = run_child(tests, 0)
    assert rc == 0, text


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mask,what", [(128, "user rows"), (256, "item gradients")])
def test_one_percent_error_is_caught_by_every_selected_parity_test(dev_lib, mask, what):
    tests = FULL_SIZE + ORACLE
    rc, text = run_child(tests, mask)
    got = outcomes(text, tests)
    assert rc != 0 and all(v is not None for v in got.values()), (what, got, text[-3000:])
    # ... and each one on an ASSERTION (about the update -- update error / err_P / err_Q / err_G / rel_err / delta_err -- or about
    # the loss of a later step, which the wrong tables of the step before have moved), not by crashing
    for t, why in got.items():
        assert "assert" in why.lower(), (t, why)


# ---- the kernels beside the step: the same question for scoring / top-k, the graph product, Adam and the pointwise branch ------
P_ = "tests/test_gpu_parity.py::"
OTHER = [
    # (variable, mask, the fault, parity tests that must each fail)
    ("RSX_SCORE_ABLATION", 8, "the product drops its last K chunk (models/MF.py:109-112)",
     [P_ + "test_score_mask_topk_match_reference_golden[g1_sgd_400x250_d128_b512-g23_400x250_d128_b512]",
      P_ + "test_score_mask_topk_match_reference_golden[g1_sgd_500x300_d64_b257-g23_500x300_d64_b257]",
      P_ + "test_score_large_tile_vs_torch_fp64", P_ + "test_fused_score_topk_at_the_bench_shape"]),
    ("RSX_SCORE_ABLATION", 16, "the filter threshold of the fused path lifted by a quarter: true Top-K items are filtered out",
     # (not the 1M-item case: there the sample's K-th score sits 1.5x below the catalog's and a quarter is not enough to cut into it)
     [P_ + "test_fused_score_topk_equals_dense_path[128-65537-1500-10]", P_ + "test_fused_score_topk_equals_dense_path[64-40001-300-50]",
      P_ + "test_fused_score_topk_on_random_shapes", P_ + "test_fused_score_topk_at_the_bench_shape"]),
    ("RSX_SCORE_ABLATION", 32, "the merge does not drop seen items (models/MF.py:130)",
     [P_ + "test_fused_score_topk_equals_dense_path[64-40001-300-50]", P_ + "test_fused_score_topk_with_long_seen_rows",
      P_ + "test_fused_score_topk_at_the_bench_shape"]),
    ("RSX_SCORE_ABLATION", 64, "the row Top-K hands out the (K+1)-th best in the K-th place (func.h:12-31)",
     [P_ + "test_topk_vs_oracle_shapes[37-1000-50]", P_ + "test_topk_vs_oracle_shapes[2-100003-50]",
      P_ + "test_score_mask_topk_match_reference_golden[g1_sgd_ml100k_d32_b256-g23_ml100k_d32_b256]", P_ + "test_topk_ties_and_masked_rows"]),
    ("RSX_GRAPH_ABLATION", 1, "one segment of the propagation product dropped (models/LightGCN.py:188-197)",
     ["tests/test_lightgcn.py::test_hip_lightgcn_matches_reference_golden", "tests/test_lightgcn.py::test_hip_spmm_long_rows_vs_oracle",
      "tests/test_lightgcn.py::test_hip_lightgcn_full_size_config5_properties"]),
    ("RSX_ABLATION", 512, "Adam with the bias corrections of step t + 1 (models/MF.py:30)",
     [P_ + "test_adam_as_shipped_matches_reference_golden", "tests/test_gpu_model.py::test_model_adam_as_shipped_matches_reference_golden",
      "tests/test_lightgcn.py::test_hip_lightgcn_matches_reference_golden"]),
    ("RSX_ABLATION", 1024, "the pointwise branch with 1 % on dL/dx (models/MF.py:99-102)",
     [P_ + "test_pointwise_branch_matches_reference_golden", "tests/test_gpu_model.py::test_pointwise_model_replays_the_reference_batches"]),
]


def outcomes_by_prefix(text, tests):
    """like outcomes(), for ids given without their parameters: every parametrization of such a test must fail"""
    failed = [line[len("FAILED "):].split(" ", 1) for line in text.splitlines() if line.startswith("FAILED ")]
    passed = "passed" in text.splitlines()[-1] if text.strip() else False
    got = {}
    for t in tests:
        hits = [(f[0], f[1] if len(f) > 1 else "failed") for f in failed if f[0] == t or f[0].startswith(t + "[")]
        got[t] = hits
    return got, passed


@pytest.mark.timeout(600)
@pytest.mark.parametrize("var,mask,what,tests", OTHER, ids=[f"{o[0][4:-9].lower() or 'step'}{o[1]}" for o in OTHER])
def test_planted_faults_in_the_other_kernels_are_caught(dev_lib, var, mask, what, tests):
    rc, text = run_child(tests, mask, var)
    got, _ = outcomes_by_prefix(text, tests)
    assert rc != 0 and all(got[t] for t in tests), (what, {t: got[t] for t in tests}, text[-3000:])
    import re
    m = re.search(r"(\d+) failed(?:, (\d+) passed)?", text)
    assert m and m.group(2) is None, (what, "some parametrization passed with the fault planted", text[-3000:])
    for t in tests:
        for _, why in got[t]:
            assert "assert" in why.lower(), (t, why)


def test_dev_library_without_a_planted_fault_passes_the_other_kernels_tests(dev_lib):
    rc, text = run_child([t for o in OTHER for t in o[3] if "bench_shape" not in t and "full_size" not in t], 0)
    assert rc == 0, text[-3000:]
