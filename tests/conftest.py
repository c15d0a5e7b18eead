import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def split_batches(g):
    """yield (u, i, j) int64 batches from a g1*/g1b* fixture."""
    off = 0
    for n in g["batch_len"]:
        n = int(n)
        yield (g["u"][off:off + n].astype(np.int64), g["i"][off:off + n].astype(np.int64),
               g["j"][off:off + n].astype(np.int64))
        off += n


def rel_err(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))
                 / (np.max(np.abs(b)) + 1e-30))


def delta_err(a, a0, b, b0=None):
    """error of the UPDATE, relative to the size of the update:
    max|(a - a0) - (b - b0)| / max|b - b0|.  `rel_err(a, b) < 1e-5` alone resolves a small update
    (lr 0.05 with 1/B gradients: ~1e-3 of max|table| after 20 steps) to a percent only."""
    b0 = a0 if b0 is None else b0
    a, a0, b, b0 = (np.asarray(x, np.float64) for x in (a, a0, b, b0))
    return float(np.max(np.abs((a - a0) - (b - b0))) / (np.max(np.abs(b - b0)) + 1e-30))


# A step size that makes the update RESOLVABLE.  The hot path carries the reference's batch mean (models/MF.py:105): every
# gradient has a factor 1/B, so at lr = 0.05 and B = 10^4 .. 10^6 one step moves an entry by 1e-8 .. 1e-6 -- at or below
# the fp32 rounding of the entry itself (3e-8 for |x| in [0.25, 0.5)), and `rel_err(table) < 1e-5` (4.5e-6 absolute) or any
# absolute slack then passes a kernel that never wrote the table.  With lr = 0.05 * B the factor 1/B cancels: a step moves
# P rows by 0.05 * sigmoid(-x) * (Q[i] - Q[j]) ~ 1e-3 .. 4e-2 and Q rows by the sum of ~B/I such terms, and the bar
# `delta_err <= UPDATE_TOL` is a bar on the update itself (1 % wrong -> delta_err 1e-2, a thousand times the bar).
UPDATE_TOL = 1e-5


def resolvable_lr(batch, base=0.05):
    return float(base) * int(batch)


def assert_update(got, start, want, what="table", tol=UPDATE_TOL):
    """the update (got - start) equals the reference's (want - start) to `tol` of the largest entry of the update"""
    e = delta_err(got, start, want)
    assert e <= tol, f"{what}: update error {e:.3e} > {tol:.1e} of the update"
    return e


# fp32 storage bounds what delta_err can resolve: a table entry of magnitude ~0.4 carries ~3e-8 of
# rounding per step in the reference's own fp32 tables against updates of 1.5e-4 .. 7e-4 of absolute
# size (G1 fixtures); the C oracle, a different summation order of the same arithmetic, sits at
# 2e-5 .. 8e-5 there.  The large-lr fixtures (G1C: updates 0.3 .. 0.8 of the table) carry the 1e-5
# bound on the update instead.
DELTA_TOL_SMALL_LR = 5e-4
# (round 5: hidden_dim 256 -- oracle/gen_golden_d256.py; G1_PADDED: hidden_dim 200, which the product stores as 256 columns -- the
#  model-level tests replay it, the kernel-level ones take d in {32, 64, 128, 256} only)
G1_SGD_BIGLR = ["g1c_sgd_biglr_200x100_d32_b64", "g1c_sgd_biglr_400x250_d128_b512", "g1c_sgd_biglr_300x180_d256_b300"]
G1_SGD = ["g1_sgd_200x100_d32_b64", "g1_sgd_ml100k_d32_b256",
          "g1_sgd_500x300_d64_b257", "g1_sgd_400x250_d128_b512", "g1_sgd_300x180_d256_b300"]
G1_ADAM = ["g1b_adam_200x100_d32_b64", "g1b_adam_ml100k_d32_b256", "g1b_adam_150x90_d256_b64"]
G1_PADDED = ["g1c_sgd_biglr_260x140_d200_b200", "g1c_sgd_biglr_150x90_d1_b64", "g1c_sgd_biglr_220x130_d77_b128"]    # (d 1, 77: oracle/gen_golden_odd_dims.py)
# the reference exactly as it ships (BASELINE configs[0]): hidden_dim 50 (stored as 64 columns), Adam, its own generator's ml-100k batches
# (oracle/gen_golden_shipped_config.py) -- replayed by the oracle test and, through the model class, on the GPU
G1_ADAM_PADDED = ["g1b_adam_ml100k_d50_b256"]
G23 = [n.replace("g1_sgd", "g23") for n in G1_SGD]
# G8: the pointwise branch (models/MF.py:99-102), oracle/gen_golden_pointwise.py
G8_POINTWISE = ["g8_pointwise_ce_sgd_300x200_d32", "g8_pointwise_mse_sgd_200x150_d64", "g8_pointwise_ce_adam_250x120_d128",
                "g8_pointwise_ce_adam_generator_120x90_d32"]


def fuzz(seed, trials):
    """(rng, number of trials) of a random-shapes test.  The committed suite runs the seed and count given here; a campaign sets
    RSX_FUZZ_SEED (added to the seed) and RSX_FUZZ_TRIALS (a multiplier) -- tools/fuzz_campaign.sh, results under profiles/"""
    extra = int(os.environ.get("RSX_FUZZ_SEED", "0"))
    mult = float(os.environ.get("RSX_FUZZ_TRIALS", "1"))
    return np.random.default_rng(seed + 7919 * extra), max(1, int(round(trials * mult)))


def split_pointwise(g):
    """yield (u, i, y) batches from a g8* fixture."""
    off = 0
    for n in g["batch_len"]:
        n = int(n)
        yield g["u"][off:off + n].astype(np.int64), g["i"][off:off + n].astype(np.int64), g["y"][off:off + n].astype(np.float32)
        off += n


@pytest.fixture(scope="session", autouse=True)
def _planted_error():
    """tests/test_mutation.py re-runs selected parity tests in a child process against the DEVELOPMENT library
    (RSX_LIB=librsx_dev.so, -DRSX_ABLATE) with a 1 % error planted in the user-row update (mask 128) or in the item
    gradients (mask 256) -- or one of the faults of the other kernels (Adam, pointwise, scoring / top-k, the graph product:
    RSX_SCORE_ABLATION, RSX_GRAPH_ABLATION) -- and expects them to FAIL.  Nothing happens unless one of the variables is set."""
    import ctypes
    for env, setter in (("RSX_ABLATION", "rsx_debug_set_ablation"), ("RSX_SCORE_ABLATION", "rsx_debug_set_score_ablation"),
                        ("RSX_GRAPH_ABLATION", "rsx_debug_set_graph_ablation")):
        mask = int(os.environ.get(env, "0"))
        if mask:
            from recsys_pytorch_amd import rsx
            fn = getattr(rsx.lib(), setter)                 # only the dev build exports it
            fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
            assert fn(mask) == 0
    yield


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build(with_ref=os.path.exists("/root/reference"))
    return oracle
