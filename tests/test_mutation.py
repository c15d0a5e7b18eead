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


def run_child(tests, mask):
    env = dict(os.environ, RSX_LIB=DEV_LIB, RSX_ABLATION=str(mask), COLUMNS="2000")     # (wide: -rf lines are not truncated)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--no-header", "--tb=line", "-rf",
                        *tests], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
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


@pytest.mark.timeout(1800)
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
