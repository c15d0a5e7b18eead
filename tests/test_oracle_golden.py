"""The CPU oracle (oracle/mf_oracle.c, oracle/torch_port.py) against the golden
vectors captured from the imported reference (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import (DELTA_TOL_SMALL_LR, G1_ADAM, G1_ADAM_PADDED, G1_PADDED, G1_SGD, G1_SGD_BIGLR, G23, G8_POINTWISE, delta_err, golden, rel_err,
                      split_batches, split_pointwise)


@pytest.mark.parametrize("name", G1_SGD + G1_SGD_BIGLR + G1_ADAM + G1_PADDED + G1_ADAM_PADDED)
def test_c_oracle_step_matches_reference(oracle_mod, name):
    g = golden(name)
    m = oracle_mod.MFOracle(g["P0"], g["Q0"], optimizer=str(g["optimizer"]), lr=float(g["lr"]))
    for t, (u, i, j) in enumerate(split_batches(g)):
        if t == 0:
            gP, gQ, _ = m.grad(u, i, j)
            assert rel_err(gP, g["gP1"]) < 2e-6   # dense grads of step 1 (MF.py:67)
            assert rel_err(gQ, g["gQ1"]) < 2e-6
        loss = m.step(u, i, j)
        assert abs(loss - g["loss"][t]) < 1e-5
    # (Adam: m / sqrt(v) amplifies the last-bit differences of a gradient summed in another order -- 1.7e-6 on the d = 256 fixture)
    bar = 3e-6 if str(g["optimizer"]) == "adam" else 1e-6
    assert rel_err(m.P, g["PT"]) < bar
    assert rel_err(m.Q, g["QT"]) < bar
    # ... and on the update itself (large-lr fixtures: the update is O(1) of the table)
    tol = 1e-5 if name in G1_SGD_BIGLR + G1_PADDED else DELTA_TOL_SMALL_LR
    assert delta_err(m.P, g["P0"], g["PT"]) < tol and delta_err(m.Q, g["Q0"], g["QT"]) < tol


@pytest.mark.parametrize("name", [G1_SGD[0], G1_ADAM[0]])
def test_torch_port_matches_reference(name):
    import torch
    from oracle.torch_port import TorchMFPort
    torch.set_num_threads(1)
    g = golden(name)
    m = TorchMFPort(g["P0"], g["Q0"], optimizer=str(g["optimizer"]), lr=float(g["lr"]))
    for t, (u, i, j) in enumerate(split_batches(g)):
        assert abs(m.step(u, i, j) - g["loss"][t]) < 1e-6
    assert rel_err(m.P, g["PT"]) < 1e-6 and rel_err(m.Q, g["QT"]) < 1e-6


@pytest.mark.parametrize("g1,g23", list(zip(G1_SGD, G23)))
def test_c_oracle_score_mask_topk(oracle_mod, g1, g23):
    a, b = golden(g1), golden(g23)
    rows = b["score_rows"].astype(np.int64)
    S = oracle_mod.score(a["PT"], a["QT"], rows)
    assert rel_err(S, b["S"]) < 2e-6                     # MF.py:109-112
    U = a["PT"].shape[0]
    full = oracle_mod.score(a["PT"], a["QT"], np.arange(U))
    oracle_mod.mask_seen(full, np.arange(U), b["mask_indptr"], b["mask_indices"])  # MF.py:130
    for K in (5, 10, 50):
        got = oracle_mod.topk(full, K)
        safe = b[f"gap_{K}"] > 1e-5                      # rows whose K-th/K+1-th gap is resolvable
        assert safe.mean() > 0.9
        for r in np.nonzero(safe)[0]:
            assert set(got[r]) == set(b[f"topk_py_{K}"][r]) == set(b[f"topk_cy_{K}"][r])
        # descending order (func.h:19)
        v = np.take_along_axis(full, got.astype(np.int64), 1)
        assert np.all(v[:, :-1] >= v[:, 1:])
        # masked items never appear
        assert not np.isneginf(v).any()


def test_c_oracle_topk_vs_reference_native(oracle_mod):
    if oracle_mod.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no reference checkout)")
    rng = np.random.default_rng(0)
    S = rng.standard_normal((37, 501)).astype(np.float32)
    for K in (1, 5, 50, 501):
        a, b = oracle_mod.topk(S, K), oracle_mod.ref_topk(S, K)
        uniq = np.array([len(np.unique(r)) == len(r) for r in S])
        assert uniq.sum() >= 30
        assert np.array_equal(a[uniq], b[uniq])          # identical where no two scores tie
        assert np.array_equal(np.take_along_axis(S, a.astype(np.int64), 1),
                              np.take_along_axis(S, b.astype(np.int64), 1))
    # exact ties: sets may differ only inside the tie class -> compare values
    T = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 1], np.float32), (3, 1))
    a, b = oracle_mod.topk(T, 5), oracle_mod.ref_topk(T, 5)
    assert np.array_equal(np.take_along_axis(T, a.astype(np.int64), 1),
                          np.take_along_axis(T, b.astype(np.int64), 1))
    assert np.array_equal(a[0], [3, 7, 9, 0, 1])         # documented tie-break: lower index first


def test_c_oracle_holdout_metrics(oracle_mod):
    g, c = golden("g4_eval_ml100k"), golden("ml100k_csr")
    res = oracle_mod.holdout(g["topk10"], [5, 10], c["valid_indptr"], c["valid_indices"].astype(np.int32))
    assert np.allclose(res, g["per_user"], atol=1e-6)    # holdout.h:29-70
    mean = res.mean(0, dtype=np.float32)
    assert np.allclose(mean, g["scores_py"], atol=1e-6)  # Evaluator.evaluate dict (python backend)
    assert np.allclose(mean, g["scores_cy"], atol=1e-6)


@pytest.mark.parametrize("name", G8_POINTWISE)
def test_c_oracle_pointwise_branch_matches_reference(oracle_mod, name):
    """models/MF.py:99-102 with hparams['pointwise'] = True (ce / mse, SGD-swapped and as-shipped Adam, random batches and
    the reference PointwiseGenerator's own): first-step dense gradients, every loss, final tables"""
    g = golden(name)
    opt, lf = str(g["optimizer"]), str(g["loss_func"])
    orc = oracle_mod.MFOracle(g["P0"], g["Q0"], optimizer=opt, lr=float(g["lr"]))
    for t, (u, i, y) in enumerate(split_pointwise(g)):
        if t == 0:
            gP, gQ, _ = orc.pointwise_grad(u, i, y, lf)
            assert rel_err(gP, g["gP1"]) < 2e-6 and rel_err(gQ, g["gQ1"]) < 2e-6
        loss = orc.pointwise_step(u, i, y, lf)
        assert abs(loss - g["loss"][t]) < 1e-5 * max(1.0, abs(g["loss"][t]))
    assert delta_err(orc.P, g["P0"], g["PT"]) < 1e-4 and delta_err(orc.Q, g["Q0"], g["QT"]) < 1e-4


def test_g5_reference_sampler_quirks():
    g, c = golden("g5_pairwise_ml100k"), golden("ml100k_csr")
    # Q3: exactly one triplet per user; Q1: "positive" is uniform over all items
    assert len(g["users"]) == int(c["num_users"]) == len(np.unique(g["users"]))
    assert float(g["frac_true_positive"]) < 0.2
    # negatives are true negatives (generators.py:178-185)
    ip, ix = c["train_indptr"], c["train_indices"]
    for u, j in zip(g["users"], g["neg"]):
        assert j not in ix[ip[u]:ip[u + 1]]


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """every oracle entry point once under -fsanitize=address,undefined (CPU build only)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True)
    if r.returncode != 0 and "cannot find" in (r.stderr + r.stdout):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0 and "asan_check ok" in r.stdout, r.stdout + r.stderr
