"""GPU: the MF model class behind the reference interface, the device sampler, and the
engine on the HIP kernels.  Run with `-m gpu` on an MI355X."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import (DELTA_TOL_SMALL_LR, fuzz, G1_PADDED, G1_SGD, G1_SGD_BIGLR, GOLDEN, assert_update, delta_err, golden, rel_err, resolvable_lr,
                      split_batches)

pytestmark = pytest.mark.gpu
HP = {"hidden_dim": 32, "pointwise": False, "loss_func": "ce"}     # conf/MF.yaml keys


@pytest.fixture(scope="module")
def ml100k():
    import recsys_pytorch_amd as pkg
    return pkg.InteractionData.from_npz(os.path.join(GOLDEN, "ml100k_csr.npz"))


def test_model_replay_matches_reference_golden(ml100k):
    """MF.train_step on the reference PairwiseGenerator's own batches (config C1 shape)"""
    import recsys_pytorch_amd as pkg
    g = golden(G1_SGD[1])
    m = pkg.MF(ml100k, dict(HP, lr=float(g["lr"])), "cuda")
    m.load_tables(g["P0"], g["Q0"])
    for t, (u, i, j) in enumerate(split_batches(g)):
        loss = m.train_step(u, i, j, users_unique=True)        # one triplet per user (quirk Q3)
        assert abs(float(loss) - g["loss"][t]) < 1e-5
    assert rel_err(m.user_embedding.weight.cpu().numpy(), g["PT"]) < 1e-5
    assert rel_err(m.item_embedding.weight.cpu().numpy(), g["QT"]) < 1e-5
    assert delta_err(m.user_embedding.weight.cpu().numpy(), g["P0"], g["PT"]) < DELTA_TOL_SMALL_LR
    assert delta_err(m.item_embedding.weight.cpu().numpy(), g["Q0"], g["QT"]) < DELTA_TOL_SMALL_LR
    # loss-only entry point (process_one_batch, MF.py:99-107) and forward (MF.py:38-42)
    u, i, j = next(split_batches(g))
    m.load_tables(g["P0"], g["Q0"])
    assert abs(float(m.process_one_batch(u, i, j)) - g["loss"][0]) < 1e-5
    r = m.forward(u, i).cpu().numpy()
    assert np.allclose(r, np.sum(g["P0"][u] * g["Q0"][i], 1), atol=1e-6)
    assert rel_err(m.user_embedding.weight.cpu().numpy(), g["P0"]) == 0.0   # nothing was updated


@pytest.mark.parametrize("name", G1_SGD_BIGLR + G1_PADDED)
def test_model_replay_large_lr_resolves_the_update_to_1e5(name):
    """the model class on the large-lr fixtures (duplicate users: general path): the 20-step update
    is 0.3-0.8 of the table, so the 1e-5 bar is a 1e-5 bar on the update itself"""
    import recsys_pytorch_amd as pkg
    g = golden(name)
    U, d = g["P0"].shape
    ds = types.SimpleNamespace(num_users=U, num_items=g["Q0"].shape[0])
    m = pkg.MF(ds, dict(HP, hidden_dim=d, lr=float(g["lr"])), "cuda")
    m.load_tables(g["P0"], g["Q0"])
    for t, (u, i, j) in enumerate(split_batches(g)):
        assert abs(float(m.train_step(u, i, j)) - g["loss"][t]) < 1e-5
    P, Q = m.user_embedding.weight.cpu().numpy(), m.item_embedding.weight.cpu().numpy()
    assert rel_err(P, g["PT"]) < 1e-5 and rel_err(Q, g["QT"]) < 1e-5
    assert delta_err(P, g["P0"], g["PT"]) < 1e-5 and delta_err(Q, g["Q0"], g["QT"]) < 1e-5


def test_model_adam_as_shipped_matches_reference_golden(ml100k):
    """hparams optimizer='adam': models/MF.py:30 semantics through the model class"""
    import recsys_pytorch_amd as pkg
    g = golden("g1b_adam_ml100k_d32_b256")
    m = pkg.MF(ml100k, dict(HP, optimizer="adam"), "cuda")
    assert m.lr == 1e-3
    m.load_tables(g["P0"], g["Q0"])
    for t, (u, i, j) in enumerate(split_batches(g)):
        assert abs(float(m.train_step(u, i, j)) - g["loss"][t]) < 1e-5
    assert rel_err(m.user_embedding.weight.cpu().numpy(), g["PT"]) < 1e-5
    assert rel_err(m.item_embedding.weight.cpu().numpy(), g["QT"]) < 1e-5


def test_model_as_the_reference_ships_it(ml100k):
    """BASELINE configs[0] to the letter: conf/MF.yaml's hidden_dim 50 (stored as 64 columns), optimizer = the shipped Adam, ml-100k, the
    first 12 batches of the reference's own generator -- losses and tables against the reference's (oracle/gen_golden_shipped_config.py)"""
    import recsys_pytorch_amd as pkg
    g = golden("g1b_adam_ml100k_d50_b256")
    m = pkg.MF(ml100k, dict(HP, hidden_dim=50, optimizer="adam"), "cuda")
    assert m.lr == 1e-3 and m._P.shape[1] == 64
    m.load_tables(g["P0"], g["Q0"])
    for t, (u, i, j) in enumerate(split_batches(g)):
        assert abs(float(m.train_step(u, i, j)) - g["loss"][t]) < 1e-5
    P, Q = m.user_embedding.weight.cpu().numpy(), m.item_embedding.weight.cpu().numpy()
    assert P.shape[1] == 50 and rel_err(P, g["PT"]) < 1e-5 and rel_err(Q, g["QT"]) < 1e-5
    assert delta_err(P, g["P0"], g["PT"]) < 1e-4 and delta_err(Q, g["Q0"], g["QT"]) < 1e-4       # (Adam's bar on the update, as in G1b)
    assert float(m._P[:, 50:].abs().max()) == 0.0 and float(m._Q[:, 50:].abs().max()) == 0.0    # the pad stays zero under Adam too
    # ... and the reference's evaluation of that model (main.py:62-63): its Evaluator's score dictionary on the valid split, and its top-10
    e = golden("g4_eval_ml100k_d50")
    top = m.predict_topk(np.arange(ml100k.num_users), ml100k.valid_input, 10)
    safe = e["gap_10"] > 1e-5
    assert safe.mean() > 0.95 and all(set(top[r]) == set(e["topk10"][r]) for r in np.nonzero(safe)[0])
    scores = pkg.Evaluator(ml100k.valid_input, ml100k.valid_target, "holdout", [5, 10]).evaluate(m)
    assert sorted(scores) == sorted(str(n) for n in e["names"])
    for n, want in zip(e["names"], e["scores_py"]):
        # (a user whose 10th and 11th scores tie to 1e-5 may swap them between two summation orders: 1 / 943 of a metric per such user)
        assert abs(float(scores[str(n)]) - float(want)) <= 1e-6 + (~safe).sum() / ml100k.num_users * 0.2, (n, float(scores[str(n)]), float(want))


def test_device_reports_the_lds_the_step_kernel_reserves_against():
    """the blocked step kernel holds its residency by LDS reservation against the CU's LDS as the DEVICE reports it
    (csrc/rsx_bpr.hip: lds_for_residency; no 160 KB constant): MI355X = 160 KB, 256 CUs, 64-wide wavefronts"""
    from recsys_pytorch_amd import rsx
    info = rsx.device_info(0)
    assert info["arch"].startswith("gfx950") and info["lds_bytes_per_cu"] == 160 * 1024, info
    assert info["wavefront_size"] == 64 and info["compute_units"] == 256, info


def test_hidden_dim_256_and_padding_d200_through_the_model(oracle_mod):
    """the reference takes any hidden_dim (models/MF.py:19,23-24): 256 runs in the EPL = 8 kernels, 200 as 256 columns with a zero pad
    -- SGD steps with repeated users, the sampled fit path, scores and top-k against the oracle"""
    import recsys_pytorch_amd as pkg
    rng = np.random.default_rng(5)
    for d in (256, 200):
        U, I, B = 3000, 900, 2500
        ds = types.SimpleNamespace(num_users=U, num_items=I)
        lr = resolvable_lr(B)
        m = pkg.MF(ds, dict(HP, hidden_dim=d, lr=lr), "cuda")
        assert m._P.shape[1] == 256
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        m.load_tables(P0, Q0)
        orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
        for t in range(3):
            u = rng.integers(0, U, B) if t else rng.permutation(U)[:B]          # unique users (in-place path), then repeats
            i, j = rng.integers(0, I, B), rng.integers(0, I, B)
            l = m.train_step(u, i, j, users_unique=(t == 0))
            assert abs(float(l) - orc.step(u, i, j)) < 1e-5
        assert_update(m.user_embedding.weight.cpu().numpy(), P0, orc.P, f"P (d={d})")
        assert_update(m.item_embedding.weight.cpu().numpy(), Q0, orc.Q, f"Q (d={d})")
        if d < 256:
            assert float(m._P[:, d:].abs().max()) == 0.0 and float(m._Q[:, d:].abs().max()) == 0.0
        users = np.arange(200)
        S = m.predict_batch_users(users).cpu().numpy()
        So = orc.score(users)
        assert rel_err(S, So) < 2e-6
        top = m.predict_topk(users, None, 20)
        srt = -np.sort(-So, axis=1)
        safe = srt[:, 19] - srt[:, 20] > 1e-5
        want = oracle_mod.topk(So, 20)
        assert safe.mean() > 0.9 and all(set(top[r]) == set(want[r]) for r in np.nonzero(safe)[0])


def test_model_on_random_shapes(oracle_mod):
    """12 random (users, items, ANY hidden_dim in 1 .. 256, batch, pairwise or pointwise / ce or mse, seen-item matrix) models through the
    class a user of the reference holds: replayed batches (train_step) against the oracle at the model's own width, the zero pad still zero,
    predict_batch_users, predict (the dense matrix with -inf at eval_pos, MF.py:114-132) and predict_topk against the oracle's scores /
    mask / partial sort"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    rng, trials = fuzz(2718, 12)
    for trial in range(trials):
        d = int(rng.integers(1, 257)) if trial % 3 else int(rng.choice([1, 31, 33, 64, 129, 255, 256]))
        U, I, B = int(rng.integers(2, 1500)), int(rng.integers(2, 1200)), int(rng.integers(1, 2000))
        pointwise = trial % 4 == 3
        lf = "mse" if (pointwise and trial % 8 == 7) else "ce"
        lr = resolvable_lr(B) * 0.2
        ctx = f"trial {trial}: U={U} I={I} hidden_dim={d} B={B} pointwise={pointwise} {lf}"
        ds = types.SimpleNamespace(num_users=U, num_items=I)
        m = pkg.MF(ds, dict(HP, hidden_dim=d, lr=lr, pointwise=pointwise, loss_func=lf), "cuda")
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        m.load_tables(P0, Q0)
        orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
        for t in range(3):
            u, i = rng.integers(0, U, B), rng.integers(0, I, B)
            if pointwise:
                y = (rng.integers(0, 2, B) if lf == "ce" else rng.integers(1, 6, B)).astype(np.float32)
                want = orc.pointwise_step(u, i, y, lf)
                got = float(m.train_step(u, i, y))
            else:
                j = rng.integers(0, I, B)
                want = orc.step(u, i, j)
                got = float(m.train_step(u, i, j))
            assert abs(got - want) < 2e-5 * max(1.0, abs(want)), (ctx, t)
        assert_update(m.user_embedding.weight.cpu().numpy(), P0, orc.P, "P, " + ctx, tol=5e-5)
        assert_update(m.item_embedding.weight.cpu().numpy(), Q0, orc.Q, "Q, " + ctx, tol=5e-5)
        if d < m._P.shape[1]:
            assert float(m._P[:, d:].abs().max()) == 0.0 and float(m._Q[:, d:].abs().max()) == 0.0, ctx
        users = rng.permutation(U)[:min(U, 150)]
        S = m.predict_batch_users(users).cpu().numpy()
        So = orc.score(users)
        assert np.abs(S - So).max() <= 2e-6 * max(np.abs(So).max(), 1e-30), ctx
        seen = sp.random(U, I, density=min(1.0, 8.0 / I), format="csr", random_state=np.random.default_rng(trial), dtype=np.float32)
        seen.data[:] = 1.0
        full = m.predict(users, seen, 64)
        ref = oracle_mod.mask_seen(orc.score(np.arange(U)), np.arange(U), seen.indptr.astype(np.int64), seen.indices.astype(np.int32))
        assert np.array_equal(np.isneginf(full[users]), np.isneginf(ref[users])), ctx
        fin = np.isfinite(ref[users])
        assert not fin.any() or np.abs(full[users][fin] - ref[users][fin]).max() <= 2e-6 * max(np.abs(So).max(), 1e-30), ctx
        K = int(rng.integers(1, min(I, 50) + 1))
        top = m.predict_topk(users, seen, K)
        for r, uu in enumerate(users):
            row = ref[uu]
            kth = np.sort(row)[-K]
            assert len(set(top[r])) == K, ctx
            if np.isfinite(kth):                                # (fewer than K unseen items: the tail is arbitrary among -inf)
                assert all(row[x] >= kth - (1e-5 + 2e-6 * np.abs(So).max()) for x in top[r]), (ctx, r)


def test_hidden_dim_padding_d50(oracle_mod):
    """conf/MF.yaml ships hidden_dim 50: stored as 64 columns, the pad stays zero"""
    import recsys_pytorch_amd as pkg
    rng = np.random.default_rng(2)
    ds = types.SimpleNamespace(num_users=120, num_items=80)
    lr = resolvable_lr(77)
    m = pkg.MF(ds, dict(HP, hidden_dim=50, lr=lr), "cuda")
    P0 = (rng.standard_normal((120, 50)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((80, 50)) * 0.1).astype(np.float32)
    m.load_tables(P0, Q0)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    for _ in range(3):
        u, i, j = rng.integers(0, 120, 77), rng.integers(0, 80, 77), rng.integers(0, 80, 77)
        l = m.train_step(u, i, j)
        assert abs(float(l) - orc.step(u, i, j)) < 1e-5
    assert_update(m.user_embedding.weight.cpu().numpy(), P0, orc.P, "P")
    assert_update(m.item_embedding.weight.cpu().numpy(), Q0, orc.Q, "Q")
    assert float(m._P[:, 50:].abs().max()) == 0.0 and float(m._Q[:, 50:].abs().max()) == 0.0
    S = m.predict_batch_users(np.arange(120)).cpu().numpy()
    assert rel_err(S, orc.score(np.arange(120))) < 2e-6


def test_predict_and_predict_topk_agree_with_oracle(ml100k, oracle_mod):
    import recsys_pytorch_amd as pkg
    g, b = golden(G1_SGD[1]), golden("g23_ml100k_d32_b256")
    m = pkg.MF(ml100k, HP, "cuda")
    m.load_tables(g["PT"], g["QT"])
    users = np.arange(ml100k.num_users)
    pred = m.predict(users, ml100k.train_data, 64)             # float64 [U x I], -inf at train positives
    assert pred.dtype == np.float64 and pred.shape == ml100k.train_data.shape
    ref = oracle_mod.score(g["PT"], g["QT"], users)
    oracle_mod.mask_seen(ref, users, b["mask_indptr"], b["mask_indices"])
    assert np.array_equal(np.isneginf(pred), np.isneginf(ref))
    ok = ~np.isneginf(ref)
    assert np.max(np.abs(pred[ok] - ref[ok])) < 2e-6 * np.max(np.abs(ref[ok]))
    for K in (5, 50):
        top = m.predict_topk(users, ml100k.train_data, K, 256)
        safe = b[f"gap_{K}"] > 1e-5
        for r in np.nonzero(safe)[0]:
            assert set(top[r]) == set(b[f"topk_cy_{K}"][r])


def test_fit_with_evaluator_end_to_end(ml100k):
    """main.py:62-68 flow: Evaluator(valid_input, valid_target) + model.fit(...)"""
    import recsys_pytorch_amd as pkg
    torch.manual_seed(2020)
    ev = pkg.Evaluator(ml100k.valid_input, ml100k.valid_target, "holdout", [5, 10])
    m = pkg.MF(ml100k, dict(HP, lr=20.0), "cuda")
    with torch.no_grad():
        m._P.mul_(0.1); m._Q.mul_(0.1)
    logged = []
    logger = types.SimpleNamespace(log_metrics=lambda d, epoch: logged.append((epoch, dict(d))))
    cfg = types.SimpleNamespace(batch_size=128, num_epochs=40, verbose=0, test_from=10, test_step=10)
    before = ev.evaluate(m)
    ret = m.fit(ml100k, cfg, evaluator=ev, loggers=[logger])
    after = ret["scores"]
    assert set(after) == {"Prec@5", "Prec@10", "Recall@5", "Recall@10", "NDCG@5", "NDCG@10"}
    assert [e for e, _ in logged] == list(range(1, 41))
    assert "NDCG@10" in logged[9][1] and "NDCG@10" not in logged[0][1]
    assert logged[-1][1]["loss"] < logged[0][1]["loss"]          # BPR loss goes down
    assert after["NDCG@10"] > before["NDCG@10"] + 0.05           # and ranking quality goes up (CPU oracle run: 0.013 -> 0.19)


@pytest.mark.parametrize("name", ["g8_pointwise_ce_adam_generator_120x90_d32", "g8_pointwise_mse_sgd_200x150_d64"])
def test_pointwise_model_replays_the_reference_batches(name):
    """MF(hparams['pointwise'] = True).train_step on the reference's own (user, item, rating) batches -- those of its
    PointwiseGenerator included -- ends at the reference's tables (models/MF.py:64-68,99-102)"""
    import recsys_pytorch_amd as pkg
    from conftest import delta_err, golden, split_pointwise
    g = golden(name)
    U, d = g["P0"].shape
    ds = types.SimpleNamespace(num_users=U, num_items=g["Q0"].shape[0])
    m = pkg.MF(ds, {"hidden_dim": d, "pointwise": True, "loss_func": str(g["loss_func"]), "optimizer": str(g["optimizer"]),
                    "lr": float(g["lr"])}, "cuda")
    m.load_tables(g["P0"], g["Q0"])
    for t, (u, i, y) in enumerate(split_pointwise(g)):
        want = float(m.process_one_batch(u, i, y))                   # the loss alone (no update) ...
        loss = float(m.train_step(u, i, y))                          # ... equals the loss of the step
        assert abs(loss - g["loss"][t]) < 1e-5 * max(1.0, abs(g["loss"][t])) and abs(want - loss) < 1e-5 * max(1.0, abs(loss))
    P, Q = m.user_embedding.weight.cpu().numpy(), m.item_embedding.weight.cpu().numpy()
    assert delta_err(P, g["P0"], g["PT"]) < 1e-4 and delta_err(Q, g["Q0"], g["QT"]) < 1e-4


def test_pointwise_fit_end_to_end(ml100k):
    """hparams['pointwise'] = True through MF.fit: the reference generator's batches (batch_size interactions + one
    negative per user, data/generators.py:105-130) assembled on the device; cross entropy goes down, ranking goes up"""
    import recsys_pytorch_amd as pkg
    torch.manual_seed(2020)
    ev = pkg.Evaluator(ml100k.valid_input, ml100k.valid_target, "holdout", [10])
    m = pkg.MF(ml100k, {"hidden_dim": 32, "pointwise": True, "loss_func": "ce", "optimizer": "adam", "lr": 5e-3}, "cuda")
    with torch.no_grad():
        m._P.mul_(0.1); m._Q.mul_(0.1)
    logged = []
    logger = types.SimpleNamespace(log_metrics=lambda d, epoch: logged.append((epoch, dict(d))))
    cfg = types.SimpleNamespace(batch_size=2048, num_epochs=6, verbose=0, test_from=6, test_step=1)
    before = ev.evaluate(m)
    after = m.fit(ml100k, cfg, evaluator=ev, loggers=[logger])["scores"]
    assert [e for e, _ in logged] == list(range(1, 7))
    assert logged[-1][1]["loss"] < logged[0][1]["loss"]
    assert after["NDCG@10"] > before["NDCG@10"] + 0.03


def test_leave_one_out_evaluator_equals_the_reference():
    """Evaluator(protocol='leave_one_out') on the reference's own leave-one-out split of ml-100k and a seeded model
    (fixture G9): the HR / NDCG means the reference's Evaluator returned"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    from conftest import golden
    g = golden("g9_loo_eval_ml100k")
    U, I = int(g["num_users"]), int(g["num_items"])
    csr = lambda part: sp.csr_matrix((np.ones(len(g[part + "_indices"])), g[part + "_indices"].astype(np.int64),
                                      g[part + "_indptr"]), shape=(U, I))
    ds = pkg.InteractionData(csr("train"), csr("valid"), csr("test"))
    m = pkg.MF(ds, {"hidden_dim": 32, "pointwise": False, "loss_func": "ce"}, "cuda")
    m.load_tables(g["P0"], g["Q0"])
    ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "leave_one_out", [int(k) for k in g["ks"]])
    got = ev.evaluate(m)
    want = dict(zip([str(n) for n in g["score_names"]], g["score_values"]))
    assert set(got) == set(want) == {"HR@1", "HR@5", "HR@10", "NDCG@1", "NDCG@5", "NDCG@10"}
    for k in want:      # a near-tie at a cut-off can move one user's hit across it: 1 / 943 = 1.1e-3
        assert abs(float(got[k]) - want[k]) < 1.5e-3, (k, got[k], want[k])
    with pytest.raises(KeyError):
        pkg.Evaluator(ds.valid_input, ds.valid_target, "hold_user_out", [5])


def test_device_sampler_properties(ml100k):
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import csr_to_device
    ip, ix = csr_to_device(ml100k.train_data, "cuda")
    U, I = ml100k.num_users, ml100k.num_items
    mk = lambda n: torch.empty(n, dtype=torch.int32, device="cuda")
    u, i, j = mk(U), mk(U), mk(U)
    rsx.bpr_sample(ip, ix, I, U, 2020, 3, 0, u, i, j)
    un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
    assert list(un) == list(range(U))                            # a full pass: every user once, in id order
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    for a, b, c in zip(un, inn, jn):
        row = ixn[ipn[a]:ipn[a + 1]]
        assert b in row and c not in row and 0 <= c < I          # true positive, true negative
    u2, i2, j2 = mk(U), mk(U), mk(U)
    rsx.bpr_sample(ip, ix, I, U, 2020, 3, 0, u2, i2, j2)
    assert torch.equal(u, u2) and torch.equal(i, i2) and torch.equal(j, j2)   # deterministic
    rsx.bpr_sample(ip, ix, I, U, 2020, 4, U, u2, i2, j2)
    assert not torch.equal(i, i2)                                # another step: other positives
    # partial batches walk a keyed PERMUTATION of the users: unique inside a pass, reshuffled per pass
    h = U // 2
    a1, a2, b1 = mk(h), mk(h), mk(h)
    rsx.bpr_sample(ip, ix, I, h, 2020, 5, 0, a1, i2[:h], j2[:h])
    rsx.bpr_sample(ip, ix, I, h, 2020, 6, h, a2, i2[:h], j2[:h])
    rsx.bpr_sample(ip, ix, I, h, 2020, 7, U, b1, i2[:h], j2[:h])
    both = torch.cat([a1, a2]).cpu().numpy()
    assert len(np.unique(both)) == 2 * h and list(a1.cpu().numpy()) != sorted(a1.cpu().numpy())
    assert not torch.equal(a1, b1)                               # next pass: another permutation
    # negatives are ~uniform over non-positives: chi-square-ish sanity on 8 item buckets
    big = 200_000
    ub, ib, jb = mk(big), mk(big), mk(big)
    for s in range(0, big, U):
        n = min(U, big - s)
        rsx.bpr_sample(ip, ix, I, n, 7, s, s // U * U, ub[s:s + n], ib[s:s + n], jb[s:s + n])
    h = np.bincount(jb.cpu().numpy() * 8 // I, minlength=8)
    assert h.min() > 0.5 * big / 8


def test_sampled_triplets_replay_through_oracle(oracle_mod):
    """parity of the sampled path: dump the device-sampled triplets, replay on the CPU oracle"""
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    U, I, d, B = 20_000, 3_000, 128, 4096
    ip, ix = synthetic_csr(U, I, 20, "cuda", seed=1)
    torch.manual_seed(3)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(I, d, device="cuda") * 0.1
    P0, Q0 = P.cpu().numpy(), Q.cpu().numpy()
    lr = resolvable_lr(B)                       # the update itself is what is compared (conftest.UPDATE_TOL)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    eng = BPREngine(P, Q, lr)
    for _ in range(6):
        u, i, j = eng.sample(ip, ix, B)
        lo = orc.step(u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy())
        acc = eng.step(u, i, j, users_unique=True)
        assert abs(float(acc.sum()) / B - lo) < 1e-5
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")


@pytest.mark.parametrize("replicas", [1, 4, 16])
def test_hot_item_replicas_do_not_change_the_sums(oracle_mod, replicas):
    """popular items' gradients spread over private replicas and folded back: same result"""
    from recsys_pytorch_amd.sharded import BPREngine
    rng = np.random.default_rng(11)
    U, I, d, B = 6000, 400, 128, 5000
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    w = 1.0 / np.arange(1, I + 1); w /= w.sum()
    lr = resolvable_lr(B)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    P, Q = torch.from_numpy(P0).cuda(), torch.from_numpy(Q0).cuda()
    eng = BPREngine(P, Q, lr)
    eng.set_hot_items(torch.from_numpy(w), num_hot=37, replicas=replicas)
    for unique in (True, False):
        for _ in range(3):
            u = rng.permutation(U)[:B] if unique else rng.integers(0, U, B)
            i, j = rng.choice(I, B, p=w), rng.integers(0, I, B)      # Zipf positives: heavy hot-row traffic
            lo = orc.step(u, i, j)
            acc = eng.step(*(torch.from_numpy(a).int().cuda() for a in (u, i, j)), users_unique=unique)
            assert abs(float(acc.sum()) / B - lo) < 1e-5
    assert float(eng.hot.ghot.abs().max()) == 0.0                    # folded and re-zeroed
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")


@pytest.mark.parametrize("cdf", [True, False])
@pytest.mark.parametrize("d,B,I,c", [(128, 40_000, 5_000, 8), (64, 30_001, 2_999, 16), (32, 9_000, 4_000, 3),
                                     (128, 50_000, 70_001, 8), (32, 20_000, 140_000, 8)])
def test_sorted_blocked_sampled_path_replays_through_oracle(oracle_mod, d, B, I, c, cdf):
    """batch sorted by positive item + negatives stratified by item block + on-chip summation:
    dump the sampled triplets, replay them on the CPU oracle, and check the sampler's guarantees"""
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    U = 50_000
    ip, ix = synthetic_csr(U, I, 12, "cuda", seed=4)
    torch.manual_seed(8)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(I, d, device="cuda") * 0.1
    P0, Q0 = P.cpu().numpy(), Q.cpu().numpy()
    lr = resolvable_lr(B)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    eng = BPREngine(P, Q, lr)
    eng.neg_block = c
    eng.use_item_cdf = cdf          # item-CDF buckets, or the device radix sort
    hist = np.zeros(I)
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    keys = set()
    for s in range(3):
        u, i, j = eng.sample(ip, ix, B)
        keys.add(eng.last_neg_key)
        un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
        assert len(np.unique(un)) == B                                   # users unique in the batch
        assert np.all(np.diff(inn) >= 0)                                 # ordered by positive item
        nominal = (np.arange(B, dtype=np.int64) * I // B) // c           # batch-position block
        bad = 0
        for w in np.unique(nominal)[:: max(1, len(np.unique(nominal)) // 200)]:
            blocks = jn[nominal == w] // c                               # all its negatives: ONE item block
            vals, cnt = np.unique(blocks, return_counts=True)
            bad += cnt.sum() - cnt.max()
        assert bad <= 0.01 * B
        for a, b_, c_ in list(zip(un, inn, jn))[:: max(1, B // 500)]:
            row = ixn[ipn[a]:ipn[a + 1]]
            assert b_ in row and c_ not in row                           # true positive, true negative
        hist += np.bincount(jn, minlength=I)
        lo = orc.step(un, inn, jn)
        acc = eng.step(u, i, j, users_unique=True, neg_block=c, neg_key=eng.last_neg_key)
        assert abs(float(acc.sum()) / B - lo) < 1e-5
    assert len(keys) == 3 and 0 not in keys                              # fresh block permutation per step
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")
    exp = 3 * B / I
    assert abs(hist.mean() - exp) < 1e-9 and hist.std() < 1.5 * np.sqrt(exp) + 1   # ~Poisson spread
    assert hist.min() > 0 or exp < 8


def _sample_sorted(ip, ix, I, B, step, cdf, sig=None, c=8, key=77, epoch_pos=0):
    from recsys_pytorch_amd import rsx
    u, i, j = (torch.full((B,), -7, dtype=torch.int32, device="cuda") for _ in range(3))
    ws = torch.empty(rsx.bpr_sample_workspace(B, I), dtype=torch.uint8, device="cuda")
    rsx.bpr_sample(ip, ix, I, B, 11, step, epoch_pos, u, i, j, neg_block=c, neg_key=key, sort_pos=True, ws=ws,
                   user_sig=sig, item_cdf=cdf)
    torch.cuda.synchronize()
    return u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()


@pytest.mark.parametrize("U,I,deg,B,pop", [(30_000, 5_000, 12, 30_000, "zipf"), (30_000, 5_000, 12, 7_001, "zipf"),
                                           (20_000, 200_000, 6, 20_000, "uniform"), (5_000, 300, 40, 5_000, "zipf"),
                                           (1_000, 50, 3, 37, "zipf")])
def test_item_cdf_buckets_order_the_same_pairs_as_the_device_sort(U, I, deg, B, pop):
    """same (seed, step) -> the bucketed layout holds exactly the (user, positive) pairs of the
    radix-sorted layout, ordered by item; twice the same call gives the same bits"""
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    ip, ix = synthetic_csr(U, I, deg, "cuda", seed=3, popularity=pop)
    cdf = rsx.build_item_cdf(ip, ix, I)
    cn = cdf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert cn[0] == 0 and cn[-1] == 0xFFFFFFFF and np.all(np.diff(cn) >= 0)
    # CDF against numpy: item mass = sum over its users of 1/deg(u)
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    mass = np.bincount(ixn, weights=np.repeat(1.0 / np.diff(ipn), np.diff(ipn)), minlength=I)
    want = np.concatenate([[0.0], np.cumsum(mass)]) / mass.sum()
    assert np.abs(cn / 2.0**32 - want).max() < 1e-6
    sig = rsx.build_signature(ip, ix, 8)
    for step in (1, 2):
        ua, ia, ja = _sample_sorted(ip, ix, I, B, step, cdf, sig)
        ub, ib, jb = _sample_sorted(ip, ix, I, B, step, None, sig)
        key_a = ia.astype(np.int64) << 32 | ua
        # ordered by item; by user inside an item's bucket (a popular item owns several buckets)
        assert np.all(np.diff(ia) >= 0) and len(np.unique(key_a)) == B
        assert np.array_equal(np.sort(key_a), np.sort(ib.astype(np.int64) << 32 | ub))
        for u_, i_, j_ in list(zip(ua, ia, ja))[:: max(1, B // 400)]:
            row = ixn[ipn[u_]:ipn[u_ + 1]]
            assert i_ in row and j_ not in row and 0 <= j_ < I
        u2, i2, j2 = _sample_sorted(ip, ix, I, B, step, cdf, sig)
        assert np.array_equal(ua, u2) and np.array_equal(ia, i2) and np.array_equal(ja, j2)
        u3, i3, j3 = _sample_sorted(ip, ix, I, B, step, cdf, None)       # signatures only skip row reads
        assert np.array_equal(ja, j3)


def test_item_cdf_buckets_on_random_shapes():
    """40 random CSRs (empty, short, long and full rows; tiny and ragged sizes; any neg_block):
    bucket layout vs radix layout hold the same pairs, ordered by item, negatives valid and inside
    the position's item block, same bits when repeated, also through the out-of-LDS path"""
    import scipy.sparse as sp
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import csr_to_device
    rng, trials = fuzz(123, 40)
    try:
        for trial in range(trials):
            U = int(rng.integers(1, 4000))
            I = int(rng.integers(2, 3000))
            kind = trial % 4
            if kind == 0:
                degs = rng.integers(0, min(I, 6), U)                       # short rows, some empty
            elif kind == 1:
                degs = rng.integers(0, min(I, 120), U)                     # beyond the 24-item register test
            elif kind == 2:
                degs = np.where(rng.random(U) < 0.05, I, rng.integers(1, min(I, 30) + 1, U))   # some own everything
            else:
                degs = np.minimum(I - 1, (rng.pareto(1.0, U) * 3).astype(np.int64))            # heavy tail
            pop = 1.0 / (1.0 + np.arange(I)) if trial % 2 else np.ones(I)
            pop = pop / pop.sum()
            rows = [np.sort(rng.choice(I, int(g), replace=False, p=pop)) for g in degs]
            indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
            indices = (np.concatenate(rows) if indptr[-1] else np.zeros(0)).astype(np.int32)
            ip = torch.from_numpy(indptr).cuda()
            ix = torch.from_numpy(indices if len(indices) else np.zeros(1, np.int32)).cuda()
            c = int(rng.integers(1, 17))
            B = U if trial % 3 == 0 else int(rng.integers(1, U + 1))
            epoch_pos = 0 if B == U else int(rng.integers(0, U - B + 1))
            cdf = rsx.build_item_cdf(ip, ix, I)
            sig = rsx.build_signature(ip, ix, c) if trial % 2 else None
            rsx.set_option("sample_sort_cap", 16 if trial % 5 == 0 else 0)
            a = _sample_sorted(ip, ix, I, B, trial, cdf, sig, c=c, key=trial * 2 + 1, epoch_pos=epoch_pos)
            b = _sample_sorted(ip, ix, I, B, trial, None, sig, c=c, key=trial * 2 + 1, epoch_pos=epoch_pos)
            a2 = _sample_sorted(ip, ix, I, B, trial, cdf, sig, c=c, key=trial * 2 + 1, epoch_pos=epoch_pos)
            ctx = f"trial {trial}: U={U} I={I} B={B} c={c}"
            for x, y in zip(a, a2):
                live2 = a[1] >= 0
                assert np.array_equal(x[live2], y[live2]), ctx
            ua, ia, ja = a
            live = ia >= 0
            dead_users = (degs == 0) | (degs >= I)
            assert len(np.unique(ua)) == B and (~live).sum() == dead_users[ua].sum(), ctx
            assert np.all(live[:live.sum()]) and np.all(ja[~live] == -1), ctx       # dead rows last
            assert np.all(np.diff(ia[live]) >= 0), ctx
            key = lambda u_, i_: np.sort(i_.astype(np.int64) << 32 | u_)
            assert np.array_equal(key(ua[live], ia[live]), key(b[0][b[1] >= 0], b[1][b[1] >= 0])), ctx
            nb = -(-I // c)
            for p in np.flatnonzero(live)[:: max(1, B // 300)]:
                row = indices[indptr[ua[p]]:indptr[ua[p] + 1]]
                assert ia[p] in row and ja[p] not in row and 0 <= ja[p] < I, ctx
            # negatives of one position range share ONE item block E; a position whose negative lies elsewhere belongs to a user who owns
            # (nearly) all of E (rsx_sample.hip draw_negative: 64 rejected draws, then the whole catalog -- owning 3/4 of E or less, that
            # has probability (3/4)^64 = 1e-8).  Per range: some block E explains every position.  (Until round 5 this was a 20 % allowance -- tools/fuzz_campaign.sh found small catalogs with long rows
            # where more users than that own their block)
            owns = np.zeros((U, I), dtype=bool)
            owns[np.repeat(np.arange(U), np.diff(indptr)), indices] = True
            w = (np.arange(B, dtype=np.int64) * I // B) // c
            blocks = ja // c
            for ww in np.unique(w[live]):
                pos = np.flatnonzero(live & (w == ww))
                seen = np.unique(blocks[pos])
                explains = lambda E: all(blocks[q] == E or owns[ua[q], E * c:min(I, (E + 1) * c)].mean() > 0.75 for q in pos)
                assert any(explains(E) for E in seen) or any(explains(E) for E in range(nb)), (ctx, int(ww), seen[:8])
            assert nb >= 1
    finally:
        rsx.set_option("sample_sort_cap", 0)


def test_item_cdf_buckets_outside_lds_and_rows_without_a_positive():
    """buckets larger than the LDS sort capacity take the in-place path; users with an empty row
    (or owning the whole catalog) come last with i = j = -1"""
    import scipy.sparse as sp
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import csr_to_device
    rng = np.random.default_rng(5)
    U, I = 12_000, 40
    dense = rng.random((U, I)) < 0.2
    dense[::7] = False                   # empty rows
    dense[3::1001] = True                # rows owning every item: no negative exists
    ip, ix = csr_to_device(sp.csr_matrix(dense.astype(np.float32)), "cuda")
    cdf = rsx.build_item_cdf(ip, ix, I)
    n_dead = int((dense.sum(1) == 0).sum() + (dense.sum(1) == I).sum())
    try:
        for cap in (2048, 64):
            rsx.set_option("sample_sort_cap", cap)
            u, i, j = _sample_sorted(ip, ix, I, U, 4, cdf, None, c=4)
            live = i >= 0
            assert (~live).sum() == n_dead and np.all(live[:U - n_dead]) and np.all(j[~live] == -1)
            assert len(np.unique(u)) == U
            assert np.all(np.diff(i[live]) >= 0)
            assert dense[u[live], i[live]].all() and not dense[u[live], j[live]].any()
            if cap == 2048:
                first = (u.copy(), i.copy(), j.copy())
            else:                                                        # same order from both paths
                assert np.array_equal(first[1], i) and np.array_equal(first[0][live], u[live]) and np.array_equal(first[2], j)
    finally:
        rsx.set_option("sample_sort_cap", 0)


def test_item_cdf_buckets_piecewise_above_two_million_positions():
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    U, I, B = 2_300_000, 20_000, 2_300_000
    ip, ix = synthetic_csr(U, I, 4, "cuda", seed=6)
    cdf = rsx.build_item_cdf(ip, ix, I)
    u, i, j = _sample_sorted(ip, ix, I, B, 1, cdf, None)
    assert len(np.unique(u)) == B and i.min() >= 0 and j.min() >= 0
    piece = 1 << 21
    for lo in (0, piece):
        assert np.all(np.diff(i[lo:lo + piece]) >= 0)                    # each piece ordered on its own
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    for q in range(0, B, 9973):
        row = ixn[ipn[u[q]]:ipn[u[q] + 1]]
        assert i[q] in row and j[q] not in row


def _full_size_step_check(U, I, d, B, deg, want_neg_block, hot=True, force_iid=False, native=False):
    """one sampled step at a BASELINE shape, verified in fp64 on the device (tests/stepcheck.py) with a step size that
    makes the update resolvable (lr = 0.05 * B, conftest.resolvable_lr): the user-row update against the gather formula,
    EVERY row; G against a dense fp64 index_add; Q after the apply; the loss -- each to 1e-5 OF THE UPDATE, no absolute
    slack.  Plus what holds at any size: every user at most once, (sorted layouts) batch ordered by positive item, true
    positives / negatives on a sample, negatives spread evenly over the catalog, rows of other users untouched.
    The step runs with the flags bench.py's native loop passes at this shape (neg_block / RSX_BATCH_SORTED from
    BPREngine._sorts), or -- native=True -- THROUGH that loop (rsx_bpr_trainer_run + rsx_bpr_trainer_last_batch)."""
    from conftest import UPDATE_TOL, resolvable_lr
    from stepcheck import verify_step
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    lr = resolvable_lr(B)
    ip, ix = synthetic_csr(U, I, deg, "cuda", seed=2020)
    torch.manual_seed(1)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(I, d, device="cuda") * 0.1
    P0, Q0 = P.clone(), Q.clone()
    eng = BPREngine(P, Q, lr)
    nb = 0 if force_iid else eng.set_neg_block(B, 8)
    assert (nb > 0) == (want_neg_block > 0) and nb <= 8, (nb, want_neg_block)   # which step path engages at this shape
    ordered = eng._sorts(B)                      # blocked kernel (nb > 0), its TILE = false form (ordered, nb == 0) or the plain kernel
    if hot:
        eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256)
    loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
    G_before_apply = None
    if native:
        tr = eng.native_trainer(ip, ix, B, loss_acc=loss)
        tr.run(1)
        torch.cuda.synchronize()
        u, i, j, nb_ran, key = tr.last_batch()
        assert nb_ran == nb and (key != 0) == (nb > 0)
        assert float(eng.G.abs().max()) == 0.0                              # the loop applied and re-zeroed G
    else:
        u, i, j = eng.sample(ip, ix, B)
        rsx.bpr_step(P, Q, eng.G, u, i, j, lr, 1.0 / B, loss_acc=loss, users_unique=True, hot=eng.hot,
                     neg_block=nb, neg_key=eng.last_neg_key, batch_sorted=ordered and not nb)
        if hot:
            rsx.fold_hot_grad(eng.G, eng.hot)
        G_before_apply = eng.G.clone()
        assert bool((Q == Q0).all())                                        # the step leaves Q alone
        rsx.apply_item_grad(Q, eng.G, lr)
        torch.cuda.synchronize()
        assert float(eng.G.abs().max()) == 0.0
    ul, il, jl = u.long(), i.long(), j.long()
    assert int(torch.bincount(ul, minlength=U).max()) == 1
    assert int(il.min()) >= 0 and int(jl.min()) >= 0 and int(jl.max()) < I
    if ordered:
        assert bool((il[1:] >= il[:-1]).all())
    pick = torch.arange(0, B, 997, device="cuda")
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    for uu, ii, jj in zip(ul[pick].tolist(), il[pick].tolist(), jl[pick].tolist()):
        row = ixn[ipn[uu]:ipn[uu + 1]]
        assert ii in row and jj not in row
    neg_hist = torch.bincount(jl, minlength=I).double()
    assert abs(float(neg_hist.mean()) - B / I) < 1e-9 and float(neg_hist.std()) < 1.5 * (B / I) ** 0.5 + 1
    r = verify_step(P0, Q0, P, Q, u, i, j, lr, 1.0 / B, G=G_before_apply)
    ctx = {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in r.items()}
    assert abs(float(loss.double().sum()) / B - r["loss"]) < 1e-5, ctx
    assert r["max_dP"] > 1e-3 and r["max_dQ"] > 1e-3, ctx                   # the updates ARE resolvable at this step size
    assert r["err_P"] <= UPDATE_TOL and r["err_Q"] <= UPDATE_TOL and r["untouched_rows_equal"], ctx
    if G_before_apply is not None:
        assert r["err_G"] <= UPDATE_TOL, ctx
        col = G_before_apply.double().sum(0)                                # each triplet adds +g p to i and -g p to j
        assert float(col.abs().max()) < 1e-6 * float(G_before_apply.double().abs().sum(0).max()) + 1e-9
    assert bool(torch.isfinite(P).all()) and bool(torch.isfinite(Q).all())


@pytest.mark.parametrize("d", [64, 128])
def test_full_size_sampled_step_invariants(d):
    """BASELINE configs[1] (d=64) and configs[2] (d=128): 1M users x 100K items, B = 1M; the bucket
    sampler + blocked kernel (bpr_step_blocked_kernel<TILE = true>) + hot-item replicas engage (neg_block 8)"""
    _full_size_step_check(1_000_000, 100_000, d, 1_000_000, 20, want_neg_block=8)


def test_full_size_config4_one_rank_slice():
    """BASELINE configs[3] as ONE of its 8 ranks sees it: 1.25M users x 1M items, d=128, 10 positives
    per user, B = 1.25M.  Q and G are 512 MB each, P 640 MB.  Here B < 2 I: fewer than two updates per item row and
    step, so the negatives are not blocked; the batch is still ordered by positive item (B >= 2^19) and the step is
    bpr_step_blocked_kernel<TILE = false> -- the kernel bench.py times at this shape."""
    _full_size_step_check(1_250_000, 1_000_000, 128, 1_250_000, 10, want_neg_block=0)


def test_full_size_independent_uniform_negatives():
    """the headline tables with independent uniform negatives (neg_block 0): batch ordered by positive item,
    bpr_step_blocked_kernel<TILE = false> (bench.py: legs.independent_uniform_negatives)"""
    _full_size_step_check(1_000_000, 100_000, 128, 1_000_000, 20, want_neg_block=0, force_iid=True)


def test_base_batch_65536_on_the_headline_tables():
    """SURVEY section 8d's base batch on the configs[2] tables: B = 65 536 < I, plain bpr_step_kernel"""
    _full_size_step_check(1_000_000, 100_000, 128, 65_536, 20, want_neg_block=0)


@pytest.mark.parametrize("shape", ["headline", "iid", "config3_slice", "base_batch"])
def test_full_size_step_through_the_native_loop(shape):
    """the same checks on a step driven by rsx_bpr_trainer_run -- the whole of bench.py's timed region: sampler on the
    trainer's side stream, step kernel, (replica fold in the) apply -- verified from rsx_bpr_trainer_last_batch"""
    if shape == "headline":
        _full_size_step_check(1_000_000, 100_000, 128, 1_000_000, 20, want_neg_block=8, native=True)
    elif shape == "iid":
        _full_size_step_check(1_000_000, 100_000, 128, 1_000_000, 20, want_neg_block=0, force_iid=True, native=True)
    elif shape == "config3_slice":
        _full_size_step_check(1_250_000, 1_000_000, 128, 1_250_000, 10, want_neg_block=0, native=True)
    else:
        _full_size_step_check(1_000_000, 100_000, 128, 65_536, 20, want_neg_block=0, native=True)


def test_sorted_runs_layout_replays_through_oracle(oracle_mod):
    """B below 2 triplets per item: the sampler still orders the batch by positive item (independent uniform
    negatives) and the step sums runs of equal positives in registers (RSX_BATCH_SORTED): dump, replay on the oracle"""
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    U, I, d, B = 60_000, 50_000, 128, 20_000
    ip, ix = synthetic_csr(U, I, 12, "cuda", seed=4)
    torch.manual_seed(8)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(I, d, device="cuda") * 0.1
    P0, Q0 = P.cpu().numpy(), Q.cpu().numpy()
    lr = resolvable_lr(B)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    eng = BPREngine(P, Q, lr)
    eng.sorted_min_batch = 16384                      # (default: 2 * I; lowered to take this layout at a small size)
    assert eng.set_neg_block(B, 8) == 0 and eng._sorts(B) and not eng._sorts(1000)
    eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 64, 4)
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    jh = np.zeros(I)
    for s in range(3):
        u, i, j = eng.sample(ip, ix, B)
        un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
        assert len(np.unique(un)) == B and np.all(np.diff(inn) >= 0)          # unique users, ordered by positive item
        for a, b_, c_ in list(zip(un, inn, jn))[::211]:
            row = ixn[ipn[a]:ipn[a + 1]]
            assert b_ in row and c_ not in row
        jh += np.bincount(jn, minlength=I)
        lo = orc.step(un, inn, jn)
        acc = eng.step(u, i, j, users_unique=True, batch_sorted=True)
        assert abs(float(acc.sum()) / B - lo) < 1e-5
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")
    assert jh.max() <= 12                                                     # negatives: independent uniform, not stratified


def test_overlapped_sampler_equals_inline_sampler():
    """sampling one step ahead on a second stream must not change a single bit"""
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    U, I, d, B = 60_000, 3_000, 128, 20_000
    ip, ix = synthetic_csr(U, I, 10, "cuda", seed=5)
    out = []
    for mode in ("inline", "overlapped"):
        torch.manual_seed(9)
        P = torch.randn(U, d, device="cuda") * 0.1
        Q = torch.randn(I, d, device="cuda") * 0.1
        P_init, Q_init = P.clone(), Q.clone()
        eng = BPREngine(P, Q, resolvable_lr(B))
        eng.set_neg_block(B, 8)
        assert 2 <= eng.neg_block <= 8
        losses = []
        for _ in range(7):   # crosses a pass boundary of the user permutation (3 batches per pass)
            fn = eng.sampled_step if mode == "inline" else eng.sampled_step_overlapped
            losses.append(float(fn(ip, ix, B).sum()))
        torch.cuda.synchronize()
        out.append((P.clone(), Q.clone(), losses))
    assert np.allclose(out[0][2], out[1][2], rtol=1e-6)              # same triplets; atomics reorder fp32 sums
    # fp32 atomics reorder sums between runs: the seven-step UPDATES agree to 1e-5 of their size, not bitwise
    assert_update(out[0][0].cpu().numpy(), P_init.cpu().numpy(), out[1][0].cpu().numpy(), "P")
    assert_update(out[0][1].cpu().numpy(), Q_init.cpu().numpy(), out[1][1].cpu().numpy(), "Q")


@pytest.mark.parametrize("U,I,B,hot", [(60_000, 3_000, 20_000, True), (30_000, 40_000, 8_192, False), (70_000, 40_000, 30_000, True)])
def test_native_trainer_equals_hand_driven_steps(U, I, B, hot):
    """rsx_bpr_trainer_run (C++ loop: sampler on its side stream || step -> apply) against the same
    steps driven from Python: same triplets (seed, step index, permutation position, per-step key),
    across several run() calls, a pass boundary of the user permutation and a change of batch size"""
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    d = 128
    ip, ix = synthetic_csr(U, I, 10, "cuda", seed=5)
    # (third case: 16 384 <= B < 2 I -- batch ordered by positive item, runs summed in registers, no blocked negatives)
    outs = []
    for mode in ("python", "native"):
        torch.manual_seed(9)
        P = torch.randn(U, d, device="cuda") * 0.1
        Q = torch.randn(I, d, device="cuda") * 0.1
        P_init, Q_init = P.clone(), Q.clone()
        eng = BPREngine(P, Q, resolvable_lr(B))
        nb = eng.set_neg_block(B, 8)
        assert (2 <= nb <= 8) if B >= 2 * I else nb == 0
        if (U, I, B) == (70_000, 40_000, 30_000):
            eng.sorted_min_batch = 16384                  # the ordered layout without blocked negatives
        if hot:
            eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 64, 4)
        total = 0.0
        if mode == "python":
            for n, bsz in ((7, B), (2, B // 3), (1, B)):
                for _ in range(n):
                    total += float(eng.sampled_step_overlapped(ip, ix, bsz).sum())
        else:
            acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
            tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
            tr.run(3); tr.run(4)
            tr.run(2, B // 3)
            tr.run(1, B)
            torch.cuda.synchronize()
            total = float(acc.sum())
            eng.adopt(tr)
            tr.close()
        torch.cuda.synchronize()
        # (the Python-driven engine's epoch_pos already counts the batch it sampled ahead)
        pos = eng.epoch_pos if mode == "native" else eng._bufs[eng._cur]["pos_before"]
        outs.append((P.clone(), Q.clone(), total, eng.step_count, pos))
    (Pa, Qa, la, sa, pa), (Pb, Qb, lb, sb, pb) = outs
    assert (sa, pa) == (sb, pb) and sa == 10
    assert abs(la - lb) < 1e-5 * abs(la)
    # fp32 atomics reorder sums between runs: the ten-step UPDATES agree to 1e-5 of their size, not bitwise
    assert_update(Pa.cpu().numpy(), P_init.cpu().numpy(), Pb.cpu().numpy(), "P")
    assert_update(Qa.cpu().numpy(), Q_init.cpu().numpy(), Qb.cpu().numpy(), "Q")


def test_native_trainer_steps_replay_through_the_oracle(oracle_mod):
    """what a native step consumed (rsx_bpr_trainer_last_batch) replayed on the CPU oracle; state and seek"""
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    U, I, d, B = 9_000, 2_000, 64, 4_000
    ip, ix = synthetic_csr(U, I, 8, "cuda", seed=2)
    torch.manual_seed(4)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(I, d, device="cuda") * 0.1
    P0, Q0 = P.cpu().numpy(), Q.cpu().numpy()
    lr = resolvable_lr(B)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    eng = BPREngine(P, Q, lr)
    assert 2 <= eng.set_neg_block(B, 8) <= 8
    acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
    tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
    ipn, ixn = ip.cpu().numpy(), ix.cpu().numpy()
    seen = []
    for t in range(4):
        acc.zero_()
        tr.run(1)
        torch.cuda.synchronize()
        u, i, j, nb, key = tr.last_batch()
        un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
        assert nb == eng.neg_block and key != 0 and len(np.unique(un)) == B and np.all(np.diff(inn) >= 0)
        for a, b_, c_ in list(zip(un, inn, jn))[::97]:
            row = ixn[ipn[a]:ipn[a + 1]]
            assert b_ in row and c_ not in row
        assert abs(float(acc.sum()) / B - orc.step(un, inn, jn)) < 1e-5
        seen.append(un)
        # two batches fit a pass of 9000 users; the third starts the next pass (tail of 1000 dropped)
        assert tr.state() == [(1, 4000), (2, 8000), (3, 13000), (4, 17000)][t]
    assert len(np.unique(np.concatenate(seen[:2]))) == 2 * B             # one pass: no user twice
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")
    # seek back to the start: the same batch again (counter-based sampler)
    tr.seek(0, 0)
    tr.run(1)
    torch.cuda.synchronize()
    assert np.array_equal(tr.last_batch()[0].cpu().numpy(), seen[0]) and tr.state()[0] == 1
    tr.close()
    with pytest.raises(rsx.RsxError):
        rsx.BPRTrainer(P, Q, eng.G, ip, ix, lr, batch=U + 1, seed=1, seed_key=1)        # batch > users


def test_fit_runs_twice_with_different_csrs(ml100k):
    """the static sampler tables (user signatures, item CDF) are bound to the CSR tensors of each
    fit(): a second fit() on ANOTHER interaction matrix must not see the first one's (a stale
    signature would accept positives as negatives)"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    from recsys_pytorch_amd.data import csr_to_device
    rng = np.random.default_rng(0)
    U, I = 4000, 600
    mats = []
    for _ in range(2):
        dense = rng.random((U, I)) < 0.02
        dense[np.arange(U), rng.integers(0, I, U)] = True
        mats.append(sp.csr_matrix(dense.astype(np.float32)))
    ds = [pkg.InteractionData(m) for m in mats]
    m = pkg.MF(ds[0], dict(HP, hidden_dim=32, lr=0.5), "cuda")
    cfg = types.SimpleNamespace(batch_size=2000, num_epochs=2, verbose=0, test_from=1, test_step=1)   # 2000 >= 2 * 600: sorted layout
    for k in range(2):
        m.fit(ds[k], cfg)
        eng = m._engine
        assert 2 <= eng.neg_block <= 8 and eng._csr is not None
        ip, ix = csr_to_device(mats[k], "cuda")
        assert torch.equal(eng._csr[0], ip) and torch.equal(eng._csr[1], ix)          # bound to THIS fit's CSR
        u, i, j = eng.sample(eng._csr[0], eng._csr[1], 2000)
        dense = np.asarray(mats[k].todense()) > 0
        un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
        assert dense[un, inn].all() and not dense[un, jn].any()


def test_whole_epoch_batches_keep_the_sampler_ahead(monkeypatch):
    """MF.fit with batch = users (one step per epoch: the headline shape as a user runs it): the loop does not seek a trainer that already
    is where the next epoch starts -- a seek drops the batches sampled ahead -- and trains the same model as the loop that seeks every
    epoch (the sampler is a function of (seed, step, position): data/generators.py:206-224 is the epoch it mirrors)"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(9)
    U, I = 5000, 400
    dense = rng.random((U, I)) < 0.03
    dense[np.arange(U), rng.integers(0, I, U)] = True
    ds = pkg.InteractionData(sp.csr_matrix(dense.astype(np.float32)))

    class NeverEqual(tuple):          # the trainer's true state, which the loop's comparison nevertheless finds "elsewhere": the old loop
        __eq__ = lambda self, other: False
        __ne__ = lambda self, other: True
        __hash__ = tuple.__hash__
    seeks = []
    real_seek, real_state = rsx.BPRTrainer.seek, rsx.BPRTrainer.state
    monkeypatch.setattr(rsx.BPRTrainer, "seek", lambda self, step, pos: (seeks.append((step, pos)), real_seek(self, step, pos))[1])
    tables = []
    for always_seek, batch in ((False, U), (True, U), (False, 1800), (True, 1800)):      # one step per epoch; two full batches and a short one
        monkeypatch.setattr(rsx.BPRTrainer, "state", (lambda self: NeverEqual(real_state(self))) if always_seek else real_state)
        del seeks[:]
        torch.manual_seed(21)
        m = pkg.MF(ds, dict(HP, hidden_dim=32, lr=0.05 * batch, seed=5), "cuda")
        P0, Q0 = m._P.clone(), m._Q.clone()
        m.fit(ds, types.SimpleNamespace(batch_size=batch, num_epochs=6, verbose=0, test_from=1, test_step=1))
        spe = -(-U // batch)
        assert seeks == ([(e * spe, e * U) for e in range(6)] if always_seek else []), (always_seek, batch, seeks)
        assert (m._engine.step_count, m._engine.epoch_pos) == (6 * spe, 6 * U)
        tables.append((m._P.clone(), m._Q.clone(), P0, Q0))
    for a, b in ((0, 1), (2, 3)):
        (Pa, Qa, P0, Q0), (Pb, Qb, P0b, Q0b) = tables[a], tables[b]
        assert torch.equal(P0, P0b) and torch.equal(Q0, Q0b)
        for x, y, z in ((Pa, Pb, P0), (Qa, Qb, Q0)):
            upd = float((x - z).abs().max())
            assert upd > 1e-3 and float((x - y).abs().max()) <= 2e-5 * upd, (a, upd, float((x - y).abs().max()))


def test_evaluation_mask_is_uploaded_once_per_matrix(monkeypatch):
    """predict_topk keeps the device copy of the seen-items matrix of the last evaluation (evaluation/evaluator.py:16-17 hands in the same
    eval_input every time): one upload for repeated evaluations, a new one for ANOTHER matrix -- also one of the same shape and size --
    and the same top-k as the dense predict + the oracle's argsort either way"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    import recsys_pytorch_amd.mf as mf_mod
    rng = np.random.default_rng(31)
    U, I = 3000, 700
    mk = lambda: sp.csr_matrix((rng.random((U, I)) < 0.02).astype(np.float32))
    A, B = mk(), mk()
    B = B[:, :].copy()
    m = pkg.MF(pkg.InteractionData(A), dict(HP, hidden_dim=32), "cuda")
    uploads = []
    real = mf_mod.csr_to_device
    monkeypatch.setattr(mf_mod, "csr_to_device", lambda mat, dev, *a, **k: (uploads.append(id(mat)), real(mat, dev, *a, **k))[1])
    users = np.arange(U)
    t1 = m.predict_topk(users, A, 10); t2 = m.predict_topk(users, A, 10)
    assert len(uploads) == 1 and np.array_equal(t1, t2)
    t3 = m.predict_topk(users, B, 10)
    assert len(uploads) == 2
    C2 = A.copy()                                        # same shape, same entries, another object: not served from the cache
    t4 = m.predict_topk(users, C2, 10)
    assert len(uploads) == 3 and np.array_equal(t4, t1)
    for top, mat in ((t1, A), (t3, B)):
        dense = m.predict(users, mat, 1024)
        want = np.argsort(-dense, axis=1, kind="stable")[:, :10]
        agree = np.mean([len(set(a) & set(b)) == 10 for a, b in zip(top, want)])
        assert agree > 0.999, agree
        seen = np.asarray(mat.todense()) > 0
        assert not seen[np.arange(U)[:, None], top].any()                    # nothing seen is recommended
    # the Evaluator's way of receiving the ids (a view of a pinned buffer the model keeps, filled by copies on a side stream): the same ids;
    # the view is only valid until the model's next such call
    for n in (U, 1, 777):
        a = m.predict_topk(users[:n], A, 10)
        b = m.predict_topk(users[:n], A, 10, reuse_host=True)
        assert b.shape == (n, 10) and b.dtype == np.int32 and np.array_equal(a, b)
    keep = b.copy()
    c = m.predict_topk(users[:777][::-1].copy(), A, 10, reuse_host=True)
    assert np.array_equal(c, keep[::-1]) and np.shares_memory(b, c)
    ev = pkg.Evaluator(A, B, "holdout", [5, 10])
    s1 = ev.evaluate(m)
    monkeypatch.setattr(type(m), "topk_reuse_host", False)
    s2 = ev.evaluate(m)
    monkeypatch.setattr(type(m), "topk_reuse_host", True)
    assert s1 == s2 and set(s1) == {"Prec@5", "Prec@10", "Recall@5", "Recall@10", "NDCG@5", "NDCG@10"}
    # the train matrix of fit likewise: a caller's own loop of one-epoch fits uploads it once (and the engine keeps its sampler tables)
    del uploads[:]
    dsA, dsB = pkg.InteractionData(A), pkg.InteractionData(B)
    cfg = types.SimpleNamespace(batch_size=U, num_epochs=1, verbose=0, test_from=1, test_step=1)
    for _ in range(3):
        m.fit(dsA, cfg)
    assert len(uploads) == 1 and m._engine.step_count == 3
    csr_a = m._engine._csr
    m.fit(dsB, cfg)
    assert len(uploads) == 2 and m._engine._csr is not csr_a
    u, i, j = m._engine.sample(m._engine._csr[0], m._engine._csr[1], U)
    seenB = np.asarray(B.todense()) > 0
    ok = np.diff(B.indptr) > 0
    un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
    live = inn >= 0
    assert seenB[un[live], inn[live]].all() and not seenB[un[live], jn[live]].any() and live.sum() == ok.sum()    # THIS fit's matrix


def test_item_block_floor_is_a_hyper_parameter():
    """hparams['neg_block_min'] (round 6): the smallest item block of the stratified negatives the engine may pick -- larger blocks mix
    the negatives of more positive items (profiles/r06_sampler_quality.txt).  Default: blocks of 3 from ten triplets per item on; the
    floor raises it up to hparams['neg_block']; every sampled negative still is a true negative of its user"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    rng = np.random.default_rng(5)
    U, I = 6000, 300
    dense = rng.random((U, I)) < 0.03
    dense[np.arange(U), rng.integers(0, I, U)] = True
    mat = sp.csr_matrix(dense.astype(np.float32))
    ds = pkg.InteractionData(mat)
    cfg = types.SimpleNamespace(batch_size=U, num_epochs=2, verbose=0, test_from=1, test_step=1)      # 20 triplets per item
    for extra, want in (({}, 3), ({"neg_block_min": 6}, 6), ({"neg_block_min": 12}, 8), ({"neg_block": 2}, 2), ({"neg_block": 16, "neg_block_min": 16}, 16)):
        m = pkg.MF(ds, dict(HP, hidden_dim=32, lr=0.5, **extra), "cuda")
        Q0 = m._Q.clone()
        m.fit(ds, cfg)
        eng = m._engine
        assert eng.neg_block == want, (extra, eng.neg_block)
        assert float((m._Q - Q0).abs().max()) > 0
        u, i, j = eng.sample(eng._csr[0], eng._csr[1], U)
        un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
        assert dense[un, inn].all() and not dense[un, jn].any() and len(np.unique(un)) == U


def test_fit_with_and_without_popular_row_replicas_trains_the_same_model():
    """MF.fit engages the replicas of the most popular items' gradient rows (hparams['hot_items'], default 256): the same
    triplets, the same sums in another order -- the tables after two epochs agree with a fit without replicas to 1e-5 of the
    update, on a popularity-skewed matrix, through the blocked kernel (batch >= 2 items) and the plain one"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    rng = np.random.default_rng(3)
    U, I = 6000, 500
    pop = 1.0 / np.arange(1, I + 1)
    pop /= pop.sum()
    rows = np.repeat(np.arange(U), 12)
    cols = rng.choice(I, size=rows.size, p=pop)
    mat = sp.csr_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(U, I))
    mat.data[:] = 1.0
    ds = pkg.InteractionData(mat)
    for batch in (3000, 400):                               # blocked path (>= 2 * 500) and the plain kernel
        cfg = types.SimpleNamespace(batch_size=batch, num_epochs=2, verbose=0, test_from=1, test_step=1)
        tables = []
        for hot in (0, 64):
            torch.manual_seed(11)                           # the tables are initialised from torch's generator, like the reference's
            m = pkg.MF(ds, dict(HP, hidden_dim=32, lr=0.05 * batch, hot_items=hot, seed=7), "cuda")
            P0, Q0 = m._P.clone(), m._Q.clone()
            m.fit(ds, cfg)
            assert (m._engine.hot is not None) == (hot > 0)
            tables.append((m._P.clone(), m._Q.clone(), P0, Q0))
        (Pa, Qa, P0, Q0), (Pb, Qb, P0b, Q0b) = tables
        assert torch.equal(P0, P0b) and torch.equal(Q0, Q0b)            # same initialisation
        for a, b, z in ((Pa, Pb, P0), (Qa, Qb, Q0)):
            upd = float((a - z).abs().max())
            assert upd > 1e-4 and float((a - b).abs().max()) <= 2e-5 * upd, (batch, upd, float((a - b).abs().max()))


@pytest.mark.timeout(900)
def test_stratified_sorted_sampler_trains_as_well_as_independent_negatives():
    """the layout the headline number rests on (batch ordered by positive item, negatives stratified by
    item block, B >= 2 I), the same as four item-range pipelines (negatives from the range of the sampled
    positive, relabelled item space), and independent uniform negatives on a planted-factor dataset: same
    model, same number of steps, two seeds each -- ranking quality (NDCG@10 on held-out positives)
    must agree within the seed-to-seed noise, and all must be far above an untrained model"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    rng = np.random.default_rng(42)
    U, I, k_true, n_pos, n_held = 40_000, 2_000, 16, 20, 5
    A, Bm = rng.standard_normal((U, k_true)), rng.standard_normal((I, k_true))
    pop = 0.8 * np.log1p(np.arange(I)[::-1])                                   # popular head, like the bench CSR
    train_rows, test_rows = [], []
    for lo in range(0, U, 4000):
        aff = A[lo:lo + 4000] @ Bm.T + pop + rng.gumbel(size=(4000, I)) * 1.5
        top = np.argpartition(-aff, n_pos, axis=1)[:, :n_pos]
        for r in top:
            r = rng.permutation(r)
            test_rows.append(np.sort(r[:n_held])); train_rows.append(np.sort(r[n_held:]))
    mk = lambda rows, n: sp.csr_matrix((np.ones(U * n), np.concatenate(rows), np.arange(U + 1) * n), shape=(U, I))
    ds = pkg.InteractionData(mk(train_rows, n_pos - n_held), mk(test_rows, n_held), mk(test_rows, n_held))
    ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "holdout", [10])
    cfg = types.SimpleNamespace(batch_size=U, num_epochs=150, verbose=0, test_from=150, test_step=150)   # B = U = 20 I
    res = {}
    for arm, nb, chunks in (("blocked", 8, 0), ("iid", 0, 0), ("ranges", 8, 4)):
        for seed in (1, 2):
            torch.manual_seed(seed)
            m = pkg.MF(ds, dict(HP, hidden_dim=32, lr=0.1 * U, neg_block=nb, chunks=chunks, seed=seed), "cuda")
            with torch.no_grad():
                m._P.mul_(0.1); m._Q.mul_(0.1)
            if (arm, seed) == ("blocked", 1):
                untrained = ev.evaluate(m)["NDCG@10"]
            out = m.fit(ds, cfg, evaluator=ev)["scores"]["NDCG@10"]
            assert (m._engine.neg_block > 0) == (nb > 0) and m._engine.chunks == chunks      # the layout under test really ran
            res[(arm, seed)] = float(out)
    a, b, c = ([res[(arm, 1)], res[(arm, 2)]] for arm in ("blocked", "iid", "ranges"))
    noise = max(abs(a[0] - a[1]), abs(b[0] - b[1]), abs(c[0] - c[1]))
    assert min(a + b + c) > max(5 * untrained, 0.05), (res, untrained)          # all learn the planted structure
    assert abs(np.mean(a) - np.mean(b)) < max(3 * noise, 0.03 * np.mean(b)), (res, untrained)
    # ... and so does the range-restricted negative sampling of the item-range pipelines
    assert abs(np.mean(c) - np.mean(b)) < max(3 * noise, 0.03 * np.mean(b)), (res, untrained)


@pytest.mark.timeout(900)
def test_item_ranges_rank_as_well_for_users_with_one_to_three_positives():
    """the item ranges pair a positive only with negatives of ITS range.  With one relabelling for a whole fit a user whose one
    to three positives fall into k < C ranges would never meet the other (C - k) / C of the catalog as negatives -- nothing would
    push those items below the user's positives in the FULL-catalog ranking; the engine therefore redraws the relabelling every
    `redraw_ranges_every` steps (BPREngine.adopt).  Planted-factor data, every user 1..3 train positives + 2 held out; Recall@20
    and NDCG@20 over the whole catalog: two ranges with the redraw == independent negatives within the seed noise (and the arm
    with ONE relabelling for the whole fit is reported beside them)"""
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    rng = np.random.default_rng(7)
    U, I, k_true = 60_000, 1_500, 12
    A, Bm = rng.standard_normal((U, k_true)), rng.standard_normal((I, k_true))
    n_train = rng.integers(1, 4, U)                                             # 1..3 train positives
    tr_r, tr_c, te_r, te_c = [], [], [], []
    for lo in range(0, U, 6000):
        aff = A[lo:lo + 6000] @ Bm.T + rng.gumbel(size=(6000, I)) * 1.0
        top = np.argpartition(-aff, 5, axis=1)[:, :5]
        for q, r in enumerate(top):
            r = rng.permutation(r)
            n = int(n_train[lo + q])
            te_r += [lo + q] * 2; te_c += list(r[:2])
            tr_r += [lo + q] * n; tr_c += list(r[2:2 + n])
    mk = lambda rr, cc: sp.csr_matrix((np.ones(len(rr)), (rr, cc)), shape=(U, I))
    train, test = mk(tr_r, tr_c), mk(te_r, te_c)
    train.sort_indices(); test.sort_indices()
    ds = pkg.InteractionData(train, test, test)
    ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "holdout", [20])
    cfg = types.SimpleNamespace(batch_size=U, num_epochs=240, verbose=0, test_from=240, test_step=240)   # B = U = 40 I
    res = {}
    for arm, nb, chunks, redraw in (("iid", 0, 0, 0), ("ranges, redrawn", 8, 2, 30), ("ranges, one relabelling", 8, 2, 0)):
        for seed in (1, 2):
            torch.manual_seed(seed)
            m = pkg.MF(ds, dict(HP, hidden_dim=32, lr=0.1 * U, neg_block=nb, chunks=chunks, seed=seed, redraw_ranges_every=redraw), "cuda")
            with torch.no_grad():
                m._P.mul_(0.1); m._Q.mul_(0.1)
            if (arm, seed) == ("iid", 1):
                untrained = ev.evaluate(m)
            members = {}                                                        # relabelling round -> the range of every item
            if chunks:
                build = m._engine._build_relabel

                def recording(ip, ix, build=build, members=members):
                    r = build(ip, ix)
                    members.setdefault(r["round"], (r["item_rank"] // r["Ic"]).cpu().numpy())
                    return r
                m._engine._build_relabel = recording
            sc = m.fit(ds, cfg, evaluator=ev)["scores"]
            assert m._engine.chunks == chunks and (m._engine._relabel_round > 0) == (redraw > 0)
            if redraw:
                # the redraw really is another partition -- I = 1500 <= 4096: every item is "heavy" here, and round 4's greedy deal
                # gave the identical membership every round (sharded.deal_items_to_ranges)
                rounds = [members[k] for k in sorted(members)]
                assert len(rounds) >= 5, sorted(members)
                for a, b in zip(rounds[:-1], rounds[1:]):
                    assert np.array_equal(np.bincount(a, minlength=chunks), np.bincount(b, minlength=chunks))
                    assert 0.3 < np.mean(a != b) < 0.7, np.mean(a != b)         # two ranges: about half of the items change sides
            elif chunks:
                assert sorted(members) == [0]
            res[(arm, seed)] = (float(sc["Recall@20"]), float(sc["NDCG@20"]))
    print({k: tuple(round(x, 4) for x in v) for k, v in res.items()}, "untrained", {k: round(float(v), 4) for k, v in untrained.items()})
    for metric in (0, 1):
        a, b = ([res[(arm, 1)][metric], res[(arm, 2)][metric]] for arm in ("iid", "ranges, redrawn"))
        noise = max(abs(a[0] - a[1]), abs(b[0] - b[1]))
        assert min(a + b) > 3 * float(untrained["Recall@20" if metric == 0 else "NDCG@20"]), (res, untrained)
        assert abs(np.mean(a) - np.mean(b)) < max(3 * noise, 0.05 * np.mean(a)), (metric, res)


def test_blocked_kernel_is_exact_on_foreign_triplets(oracle_mod):
    """neg_block set but the triplets do NOT follow the sampler contract: still exact"""
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(21)
    U, I, d, B = 9000, 777, 128, 6000
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    lr = resolvable_lr(B)
    orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    P, Q = torch.from_numpy(P0).cuda(), torch.from_numpy(Q0).cuda()
    G = torch.zeros_like(Q)
    for _ in range(3):
        u, i, j = rng.permutation(U)[:B], rng.integers(0, I, B), rng.integers(0, I, B)
        orc.step(u, i, j)
        rsx.bpr_step(P, Q, G, *(torch.from_numpy(a).int().cuda() for a in (u, i, j)), lr, 1.0 / B,
                     users_unique=True, neg_block=8)
        rsx.apply_item_grad(Q, G, lr)
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")


def test_synthetic_csr_shape_and_popularity():
    from recsys_pytorch_amd.data import synthetic_csr
    ip, ix = synthetic_csr(50_000, 10_000, 20, "cuda", seed=2020)
    rows = ix.view(50_000, 20).cpu().numpy()
    assert np.all(np.diff(rows, axis=1) > 0)                     # sorted, no duplicates per user
    cnt = np.bincount(rows.reshape(-1), minlength=10_000)
    assert cnt[0] > 20 * cnt[1000:1100].mean()                   # head is much more popular than the tail


@pytest.mark.parametrize("d,B,hot,mode", [(128, 256, 64, 1), (64, 4096, 0, 1), (128, 20000, 64, 2), (32, 1, 8, 1)])
def test_small_batches_apply_the_marked_rows_only(oracle_mod, d, B, hot, mode):
    """batches small against the catalog (the reference's default batch of 256 among them, config.py) through the native loop with the
    row-marked apply (include/rsx.h: rsx_bpr_trainer_config.touched -- the plain kernel marks the rows of G it adds to, the apply
    visits those instead of sweeping G): every step replays through the CPU oracle (loss 1e-5, the update of P and Q to 1e-5 of its
    size), popular rows with replicas included, and afterwards no mark and no gradient is left behind.  mode 2 = the marked apply
    forced where the rule (2 B <= items) would sweep; the same steps with the option off (the sweep) agree"""
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    U, I, steps = 30_000, 30_000, 5
    ip, ix = synthetic_csr(U, I, 9, "cuda", seed=11, popularity="zipf")
    lr = resolvable_lr(B)
    outs = []
    for option in (mode, 0):
        rsx.set_option("touched_apply", option)
        try:
            torch.manual_seed(6)
            P = torch.randn(U, d, device="cuda") * 0.1
            Q = torch.randn(I, d, device="cuda") * 0.1
            P0, Q0 = P.cpu().numpy().copy(), Q.cpu().numpy().copy()
            orc = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
            eng = BPREngine(P, Q, lr)
            assert eng.set_neg_block(B, 8) == 0                      # far below two triplets per item: the plain kernel
            eng.sorted_min_batch = 0
            if hot:
                eng.set_hot_items(torch.bincount(ix.long(), minlength=I), hot, 4)
            acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
            tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
            assert tr.touched is not None and tr.touched.numel() == I
            for _ in range(steps):
                acc.zero_()
                tr.run(1)
                torch.cuda.synchronize()
                u, i, j = (x.cpu().numpy().astype(np.int64) for x in tr.last_batch()[:3])
                assert abs(float(acc.sum()) / B - orc.step(u, i, j)) < 1e-5
                assert int(tr.touched.sum()) == 0 and float(eng.G.abs().max()) == 0.0       # nothing left behind
                if eng.hot is not None:
                    assert float(eng.hot.ghot.abs().max()) == 0.0
            assert_update(P.cpu().numpy(), P0, orc.P, "P")
            assert_update(Q.cpu().numpy(), Q0, orc.Q, "Q")
            tr.run(3)                                                 # ... and several steps queued by one call
            torch.cuda.synchronize()
            assert int(tr.touched.sum()) == 0 and float(eng.G.abs().max()) == 0.0
            eng.adopt(tr)
            outs.append((P.cpu().numpy(), Q.cpu().numpy()))
            tr.close()
        finally:
            rsx.set_option("touched_apply", 1)
    # the marked apply and the sweep took the same eight steps (same sampler draws): the tables agree to the atomics' summation order
    assert_update(outs[0][0], P0, outs[1][0], "P (marked apply vs sweep)")
    assert_update(outs[0][1], Q0, outs[1][1], "Q (marked apply vs sweep)")
