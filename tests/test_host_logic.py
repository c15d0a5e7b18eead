"""Host logic on the CPU: metrics behind the C ABI, shard arithmetic, the step engine
(with the oracle-backed kernels stand-in), and the loud failure without a GPU."""
import numpy as np
import pytest
import torch

import cpu_kernels
from conftest import G1_SGD, golden, rel_err, split_batches


def test_eval_holdout_matches_reference_golden():
    from recsys_pytorch_amd import rsx
    g, c = golden("g4_eval_ml100k"), golden("ml100k_csr")
    res = rsx.eval_holdout(g["topk10"], [5, 10], c["valid_indptr"], c["valid_indices"].astype(np.int32))
    assert np.allclose(res, g["per_user"], atol=1e-6)             # holdout.h:29-70
    assert np.allclose(res.mean(0, dtype=np.float32), g["scores_py"], atol=1e-6)


def test_eval_holdout_matches_reference_native(oracle_mod):
    from recsys_pytorch_amd import rsx
    if oracle_mod.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(3)
    n, I, mk = 200, 500, 20
    rk = np.stack([rng.permutation(I)[:mk] for _ in range(n)]).astype(np.int32)
    lens = rng.integers(1, 40, n)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ix = np.concatenate([rng.permutation(I)[:l] for l in lens]).astype(np.int32)   # unsorted rows
    a = rsx.eval_holdout(rk, [1, 5, 20], ip, ix)
    b = oracle_mod.holdout(rk, [1, 5, 20], ip, ix, use_ref=True)
    assert np.allclose(a, b, atol=1e-6)


def test_eval_loo_matches_reference_golden(oracle_mod):
    """rsx_eval_loo against what the reference's Evaluator(protocol='leave_one_out') accumulated per user
    (evaluation/backend/python/loo.py:11-32) on its own rankings, and against its loo.h compiled in oracle/_ref"""
    from recsys_pytorch_amd import rsx
    g = golden("g9_loo_eval_ml100k")
    ks = [int(k) for k in g["ks"]]
    truth = g["valid_indices"][g["valid_indptr"][:-1]].astype(np.int32)
    res = rsx.eval_loo(g["topk10"], ks, truth)
    assert np.allclose(res, g["per_user"], atol=1e-6)
    assert np.allclose(res, oracle_mod.loo(g["topk10"], ks, truth), atol=1e-6)
    means = dict(zip([str(n) for n in g["score_names"]], g["score_values"]))
    for m, metric in enumerate(("HR", "NDCG")):
        for q, k in enumerate(ks):
            assert abs(float(np.mean(res[:, m * len(ks) + q], dtype=np.float32)) - means["%s@%d" % (metric, k)]) < 1e-6
    if oracle_mod.ref_lib() is not None and hasattr(oracle_mod.ref_lib(), "ref_evaluate_loo"):
        rng = np.random.default_rng(4)
        rk = np.stack([rng.permutation(300)[:20] for _ in range(500)]).astype(np.int32)
        t = rng.integers(0, 300, 500).astype(np.int32)
        assert np.allclose(rsx.eval_loo(rk, [1, 7, 20], t), oracle_mod.loo(rk, [1, 7, 20], t, use_ref=True), atol=1e-6)
    with pytest.raises(rsx.RsxError):
        rsx.eval_loo(np.zeros((2, 5), np.int32), [6], np.zeros(2, np.int32))


def test_eval_metrics_on_random_shapes_match_reference_native(oracle_mod):
    """60 random (users, items, ranking length, cut-offs in any order, holdout rows of any length incl. EMPTY ones -- NaN like the
    reference's 0 / 0) problems: rsx_eval_holdout / rsx_eval_loo == the reference's holdout.h / loo.h compiled in oracle/_ref == the
    oracle's restatement"""
    from conftest import fuzz
    from recsys_pytorch_amd import rsx
    use_ref = oracle_mod.ref_lib() is not None
    rng, trials = fuzz(31, 60)
    for trial in range(trials):
        n, I = int(rng.integers(1, 300)), int(rng.integers(2, 800))
        mk = int(rng.integers(1, min(I, 60) + 1))
        rk = np.stack([rng.permutation(I)[:mk] for _ in range(n)]).astype(np.int32)
        lens = rng.integers(0 if trial % 3 == 0 else 1, min(I, 50) + 1, n)
        ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ix = (np.concatenate([rng.permutation(I)[:l] for l in lens]) if ip[-1] else np.zeros(1)).astype(np.int32)
        ks = sorted({int(x) for x in rng.integers(1, mk + 1, int(rng.integers(1, 5)))}, reverse=trial % 5 == 0)
        ctx = f"trial {trial}: n={n} I={I} mk={mk} ks={ks} shortest row {lens.min()}"
        a = rsx.eval_holdout(rk, ks, ip, ix)
        assert np.allclose(a, oracle_mod.holdout(rk, ks, ip, ix), atol=1e-6, equal_nan=True), ctx
        t = rng.integers(0, I, n).astype(np.int32)
        b = rsx.eval_loo(rk, ks, t)
        assert np.allclose(b, oracle_mod.loo(rk, ks, t), atol=1e-6), ctx
        if use_ref:
            assert np.allclose(a, oracle_mod.holdout(rk, ks, ip, ix, use_ref=True), atol=1e-6, equal_nan=True), ctx
            if hasattr(oracle_mod.ref_lib(), "ref_evaluate_loo"):
                assert np.allclose(b, oracle_mod.loo(rk, ks, t, use_ref=True), atol=1e-6), ctx


def test_eval_metrics_over_host_threads_equal_one_thread():
    """rsx_eval_holdout / rsx_eval_loo cut the users into ranges over host threads from 8 192 users on (csrc/rsx_eval.hip; holdout.h:20-103
    and loo.h:19-85 are per-user loops): every user's numbers are the ones a single thread computes -- compared with the same call over
    slices below the threshold, bit for bit, at sizes that leave a ragged last range"""
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(77)
    for n in (8_191, 8_192, 50_001, 131_072 + 5):
        K = 20
        rk = rng.integers(0, 500, (n, K)).astype(np.int32)
        deg = rng.integers(0, 9, n)
        deg[rng.integers(0, n, 50)] = 0                                       # users without targets stay in the call (evaluator.py skips them in the mean)
        ip = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        ix = rng.integers(0, 500, int(ip[-1])).astype(np.int32)
        whole = rsx.eval_holdout(rk, [1, 5, 20], ip, ix)
        parts = np.concatenate([rsx.eval_holdout(rk[s:s + 3000], [1, 5, 20], ip[s:s + 3001] - ip[s], ix[ip[s]:ip[min(s + 3000, n)]]) for s in range(0, n, 3000)])
        has = deg > 0
        assert whole.shape == (n, 9) and np.array_equal(whole[has], parts[has])
        truth = rng.integers(0, 500, n).astype(np.int32)
        whole = rsx.eval_loo(rk, [1, 5, 20], truth)
        parts = np.concatenate([rsx.eval_loo(rk[s:s + 3000], [1, 5, 20], truth[s:s + 3000]) for s in range(0, n, 3000)])
        assert np.array_equal(whole, parts)


def test_eval_holdout_rejects_bad_k():
    from recsys_pytorch_amd import rsx
    with pytest.raises(rsx.RsxError):
        rsx.eval_holdout(np.zeros((2, 5), np.int32), [6], np.array([0, 1, 2]), np.array([0, 1], np.int32))


def test_pick_neg_block_follows_the_batch_density():
    """sharded.pick_neg_block: short wavefronts (the smallest block that leaves 20 positions) from 10 triplets per item on, the
    fullest-last-round rule below, never above max_block, item ranges from 3 up"""
    from recsys_pytorch_amd.sharded import pick_neg_block
    slots = 256 * 24
    assert pick_neg_block(100_000, 8, slots, 1 << 20) == 3            # the headline shape: 10.5 triplets per item; blocks of 3 since round 6
    assert pick_neg_block(100_000, 8, slots, 1 << 20, 2) == 2         # ... (2 gave the 20 positions; 3 is free on the clock and mixes more positives)
    assert pick_neg_block(100_000, 8, slots, 1_000_000) == 3
    assert pick_neg_block(100_000, 8, slots, 262_144) == 6            # 2.6 per item: whole rounds of wavefronts
    assert pick_neg_block(100_000, 8, slots) == 6                     # no batch given: the round rule
    assert pick_neg_block(100_000, 4, slots, 1 << 20, 3) == 3
    assert pick_neg_block(100_000, 2, slots, 1 << 20, 3) == 2         # min_block never exceeds max_block
    assert pick_neg_block(100_000, 1, slots, 1 << 20) == 1
    assert pick_neg_block(1_000, 8, slots, 100_000) == 3              # 100 per item: the floor
    for I, B in ((50_000, 600_000), (7_777, 90_000), (100_000, 3_000_000), (100_000, 1_000_001)):
        c = pick_neg_block(I, 8, slots, B)
        assert 3 <= c <= 8 and c * B >= 20 * I and (c == 3 or (c - 1) * B < 20 * I)
    # the caller's floor (hparams['neg_block_min']: larger blocks mix the negatives of more positive items) bounds BOTH rules from below
    assert pick_neg_block(100_000, 8, slots, 1 << 20, floor=8) == 8 and pick_neg_block(100_000, 8, slots, 1 << 20, 3, floor=4) == 4
    assert pick_neg_block(100_000, 8, slots, 1 << 20, 3, floor=2) == 3 and pick_neg_block(100_000, 8, slots, 1 << 20, 2, floor=2) == 2 and pick_neg_block(100_000, 4, slots, 1 << 20, floor=16) == 4
    assert pick_neg_block(100_000, 16, slots, 262_144, floor=7) >= 7 and pick_neg_block(100_000, 8, slots, floor=1) == 6


def test_user_block_partition():
    from recsys_pytorch_amd.sharded import user_block
    for U, W in ((10, 3), (1_000_000, 8), (7, 8), (943, 2)):
        blocks = [user_block(U, r, W) for r in range(W)]
        assert blocks[0][0] == 0 and blocks[-1][1] == U
        for (a, b), (c, d) in zip(blocks, blocks[1:]):
            assert b == c and a <= b


@pytest.mark.parametrize("name", [G1_SGD[0], G1_SGD[2]])
def test_engine_replay_matches_reference_golden_on_cpu_kernels(name):
    """the engine's step sequence (step -> apply) with the oracle-backed kernels"""
    from recsys_pytorch_amd.sharded import BPREngine
    g = golden(name)
    P, Q = torch.from_numpy(g["P0"].copy()), torch.from_numpy(g["Q0"].copy())
    eng = BPREngine(P, Q, float(g["lr"]), kernels=cpu_kernels)
    for t, (u, i, j) in enumerate(split_batches(g)):
        acc = eng.step(torch.from_numpy(u).int(), torch.from_numpy(i).int(), torch.from_numpy(j).int())
        assert abs(float(acc.sum()) / len(u) - g["loss"][t]) < 1e-5
    assert rel_err(P.numpy(), g["PT"]) < 1e-5 and rel_err(Q.numpy(), g["QT"]) < 1e-5


def test_engine_sampler_epoch_logic():
    from recsys_pytorch_amd.sharded import BPREngine
    c = golden("ml100k_csr")
    U, I = int(c["num_users"]), int(c["num_items"])
    ip = torch.from_numpy(c["train_indptr"]); ix = torch.from_numpy(c["train_indices"].astype(np.int32))
    eng = BPREngine(torch.zeros(U, 32), torch.zeros(I, 32), 0.05, kernels=cpu_kernels)
    seen = []
    for _ in range(4):   # 943 users, batch 256: 3 full batches then a restart (tail dropped)
        u, i, j = eng.sample(ip, ix, 256)
        assert len(torch.unique(u)) == 256
        seen.append(u.clone())
    assert eng.epoch_pos == 943 + 256          # 4th batch started a new pass
    assert len(torch.unique(torch.cat(seen[:3]))) == 768


def test_model_fails_loudly_without_gpu():
    """no silent CPU path: the product model on a host device must raise, not compute"""
    import recsys_pytorch_amd as pkg
    from recsys_pytorch_amd import rsx
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ds = pkg.InteractionData.from_npz(__import__("os").path.join(__import__("conftest").GOLDEN, "ml100k_csr.npz"))
    m = pkg.MF(ds, {"hidden_dim": 32, "pointwise": False, "loss_func": "ce"}, "cpu")
    with pytest.raises(rsx.RsxError):
        m.train_step(np.array([0, 1]), np.array([1, 2]), np.array([3, 4]))
    with pytest.raises(rsx.RsxError):
        m.predict_batch_users(np.array([0, 1]))


def test_registry_and_interface_names():
    import recsys_pytorch_amd as pkg
    cls = getattr(pkg, "MF")                                   # main.py:46-47
    for name in ("forward", "fit", "predict", "embeddings", "process_one_batch", "predict_batch_users"):
        assert callable(getattr(cls, name))
    assert issubclass(cls, pkg.BaseModel) and issubclass(pkg.BaseModel, torch.nn.Module)
    # the constructor reads the reference's three hparams (models/MF.py:19-21); pointwise / loss_func select the branch
    import cpu_kernels
    m = cls(type("D", (), {"num_users": 4, "num_items": 4})(), {"hidden_dim": 8, "pointwise": True, "loss_func": "mse"}, "cpu",
            kernels=cpu_kernels)
    assert m.pointwise and m.loss_func == "mse"


def test_pointwise_fit_loop_assembles_the_reference_generators_batches():
    """MF(hparams['pointwise'] = True).fit on the CPU stand-in kernels: every batch is `batch_size` interactions of a
    per-epoch permutation PLUS one negative (rating 0, never a positive of that user) for EVERY user
    (data/generators.py:105-130,79-100), the number of batches is ceil(nnz / batch_size) (:102-103), every
    interaction is visited once per epoch, and the loss falls"""
    import types
    import recsys_pytorch_amd as pkg
    from recsys_pytorch_amd.sharded import BPREngine
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "g8_pointwise_generator_matrix.npz"))
    import scipy.sparse as sp
    U, I = (int(x) for x in g["shape"])
    vals = np.random.default_rng(2).integers(1, 6, len(g["indices"])).astype(np.float32)     # explicit ratings 1..5
    mat = sp.csr_matrix((vals, g["indices"], g["indptr"]), shape=(U, I))
    mat = mat[:, ::-1].tocsr()                 # a matrix whose rows are NOT stored in ascending item order
    mat = sp.csr_matrix((mat.data, mat.indices, mat.indptr), shape=(U, I))
    assert not mat.has_sorted_indices
    before = (mat.indices.copy(), mat.data.copy())
    ds = pkg.InteractionData(mat)
    torch.manual_seed(1)
    m = pkg.MF(ds, {"hidden_dim": 32, "pointwise": True, "loss_func": "mse", "optimizer": "adam", "lr": 0.01}, "cpu",
               kernels=cpu_kernels)
    seen = []
    orig = BPREngine.pointwise_step

    def spy(self, u, i, y, loss_func="ce", count_step=True):
        seen.append((u.clone(), i.clone(), y.clone()))
        return orig(self, u, i, y, loss_func, count_step)
    BPREngine.pointwise_step = spy
    try:
        logged = []
        logger = types.SimpleNamespace(log_metrics=lambda d, epoch: logged.append(dict(d)))
        m.fit(ds, types.SimpleNamespace(batch_size=64, num_epochs=3, verbose=0, test_from=1, test_step=1), loggers=[logger])
    finally:
        BPREngine.pointwise_step = orig
    nnz, bs = mat.nnz, 64
    per_epoch = -(-nnz // bs)
    assert len(seen) == 3 * per_epoch
    full = mat.toarray()
    dense = full > 0
    assert np.array_equal(mat.indices, before[0]) and np.array_equal(mat.data, before[1])      # the caller's matrix is untouched
    for e in range(3):
        pairs = set()
        for b, (u, i, y) in enumerate(seen[e * per_epoch:(e + 1) * per_epoch]):
            n_pos = min(bs, nnz - b * bs)
            assert len(u) == n_pos + U                                      # + one negative for EVERY user
            assert (y[n_pos:] == 0).all()
            assert np.array_equal(y[:n_pos].numpy(), np.asarray(full[u[:n_pos].numpy(), i[:n_pos].numpy()]).ravel())   # ITS rating
            assert sorted(u[n_pos:].tolist()) == list(range(U))             # every user once
            assert not dense[u[n_pos:].numpy(), i[n_pos:].numpy()].any()    # true negatives
            pairs.update(zip(u[:n_pos].tolist(), i[:n_pos].tolist()))
        assert len(pairs) == nnz                                            # the whole matrix once per epoch
    assert logged[-1]["loss"] < logged[0]["loss"]


@pytest.mark.parametrize("I,C,pop", [(100_000, 2, "zipf"), (100_000, 4, "zipf"), (1500, 2, "zipf"), (3000, 3, "uniform"), (7, 2, "zipf")])
def test_item_range_redraw_moves_heavy_items_and_keeps_the_balance(I, C, pop):
    """sharded.deal_items_to_ranges: two items meet as positive / negative only while they share an item range, so the redraw
    must change the membership of the HEAVY items too (round 4 dealt the 4096 heaviest by a greedy loop that depended on the
    masses alone: identical every round, and for I <= 4096 the whole partition was) -- while every range keeps its seat count
    and its share of the sampling mass"""
    from recsys_pytorch_amd.sharded import deal_items_to_ranges
    mass = 1.0 / (np.arange(I) + 1.0) if pop == "zipf" else np.ones(I)
    base, rem = divmod(I, C)
    cap = np.array([base + (k < rem) for k in range(C)])
    rounds = [deal_items_to_ranges(mass, cap, np.random.default_rng(2020 * 7919 + 13 + 104729 * r)) for r in range(6)]
    again = deal_items_to_ranges(mass, cap, np.random.default_rng(2020 * 7919 + 13))
    assert np.array_equal(rounds[0], again)                       # a function of (mass, seed, round): every rank draws the same
    for a in rounds:
        assert a.min() >= 0 and np.array_equal(np.bincount(a, minlength=C), cap)
        if I >= 1000:
            load = np.bincount(a, weights=mass, minlength=C)
            assert load.max() / load.mean() < 1.02, load / load.sum()
    if I < 1000:
        return
    heavy = np.argsort(-mass, kind="stable")[:min(I, 4096)]
    for a, b in zip(rounds[:-1], rounds[1:]):
        moved = np.mean(a[heavy] != b[heavy])
        assert moved > 0.8 * (1 - 1.0 / C) , moved                # membership of the heavy items is redrawn (expected: 1 - 1/C)
    # over a handful of rounds the two heaviest items share a range at least once, and any heavy pair does
    together = lambda x, y: sum(int(a[x] == a[y]) for a in rounds)
    pairs = [(heavy[k], heavy[k + 1]) for k in range(0, 64, 2)]
    assert np.mean([together(x, y) > 0 for x, y in pairs]) > 0.8


def _replay_hot_plan(plan, X, waves=16):
    """the scatter of include/rsx.h: rsx_spmm_hot_rows restated on the host: per (chunk of source rows, wavefront) a run of entries,
    ordered by the wavefront's slot (rw_off), each adding val * X[source] to its slot's accumulator; a row's result is the sum of
    the slots that name it (hot_rows)"""
    H, K = plan["num_slots"], plan["chunk_rows"]
    R = H // waves
    acc = np.zeros((H, X.shape[1]))
    rw = plan["rw_off"].astype(np.int64).reshape(-1, R + 1)
    nchunks = -(-len(plan["src_rows"]) // K)
    assert rw.shape[0] == nchunks * waves and len(plan["cw_ptr"]) == nchunks * waves + 1
    assert (np.diff(plan["cw_ptr"]) == rw[:, -1]).all() and (np.diff(rw, axis=1) >= 0).all() and (rw[:, 0] == 0).all()
    for cw in range(nchunks * waves):
        c, w = divmod(cw, waves)
        base = int(plan["cw_ptr"][cw])
        for r in range(R):
            e = np.arange(base + rw[cw, r], base + rw[cw, r + 1])
            if len(e) == 0:
                continue
            code = plan["ent_code"][e].astype(np.int64)
            assert ((code >> 8) == r).all()                       # the slot the entry carries is the slot its position says
            src = c * K + (code & 0xFF)
            assert (src < len(plan["src_rows"])).all()
            acc[w * R + r] += (plan["ent_val"][e, None].astype(np.float64) * X[plan["src_rows"][src]]).sum(0)
    return acc


@pytest.mark.parametrize("d", [32, 128, 256])
def test_hot_row_plan_covers_every_entry_of_its_rows_once(d):
    """recsys_pytorch_amd/rsx.py: spmm_hot_plan (host side of include/rsx.h: rsx_spmm_hot) on popularity-skewed graphs: the plan's rows
    are the longest ones, every slot belongs to one wavefront's share, and replaying the plan gives A[rows] @ X (models/LightGCN.py:188-197
    is the product it is part of)"""
    import scipy.sparse as sp
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(600 + d)
    for trial in range(4):
        n = int(rng.integers(300, 3000))
        nnz = int(rng.integers(n, 30 * n))
        pop = rng.zipf(1.3, nnz) % n                                   # a few very long rows
        A = sp.csr_matrix((rng.standard_normal(nnz).astype(np.float32), (pop, rng.integers(0, n, nnz))), shape=(n, n))
        A.sum_duplicates(); A.sort_indices()
        plan, share = rsx.spmm_hot_plan(A, d)
        H = plan["num_slots"]
        assert H == int(rsx.lib().rsx_spmm_hot_capacity(d)) and plan["chunk_rows"] == int(rsx.lib().rsx_spmm_hot_chunk_rows(d))
        rows = plan["uniq_rows"]
        lens = np.diff(A.indptr)
        assert len(np.unique(rows)) == len(rows) and (lens[rows] > 0).all()
        assert lens[rows].min() >= np.sort(lens)[::-1][min(H, n) - 1]   # none of them shorter than the H-th longest row
        assert abs(share - lens[rows].sum() / A.nnz) < 1e-12
        named = plan["hot_rows"][plan["hot_rows"] >= 0]
        assert set(named.tolist()) == set(rows.tolist())
        assert len(plan["ent_code"]) == len(plan["ent_val"]) == int(plan["cw_ptr"][-1]) == int(lens[rows].sum())
        assert (np.diff(plan["src_rows"]) > 0).all()
        X = rng.standard_normal((n, 3))
        acc = _replay_hot_plan(plan, X)
        got = np.zeros((n, 3))
        np.add.at(got, plan["hot_rows"][plan["hot_rows"] >= 0], acc[plan["hot_rows"] >= 0])
        want = np.zeros((n, 3)); want[rows] = A[rows].astype(np.float64) @ X
        assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), (d, trial, n, nnz)
    assert rsx.spmm_hot_plan(sp.csr_matrix((50, 50), dtype=np.float32), d) == (None, 0.0)


def test_csc_walk_rank_is_in_range_and_uniform():
    """recsys_pytorch_amd/rsx.py: csc_positive_rank (the host restatement the device walk is compared with): 0 <= rank < deg, a different
    draw every step, every rank of a row about equally often (data/generators.py:160-166 draws the positive uniformly in the row)"""
    from recsys_pytorch_amd import rsx
    users = np.arange(200_000)
    deg = (users % 7 + 1).astype(np.int64)
    r0, r1 = rsx.csc_positive_rank(users, deg, 2020, 0), rsx.csc_positive_rank(users, deg, 2020, 1)
    assert (r0 >= 0).all() and (r0 < deg).all() and (r1 < deg).all()
    assert (r0[deg > 1] != r1[deg > 1]).mean() > 0.5
    assert (rsx.csc_positive_rank(users, deg, 2020, 0) == r0).all() and (rsx.csc_positive_rank(users, deg, 2021, 0)[deg > 1] != r0[deg > 1]).any()
    for k in (2, 5, 7):
        cnt = np.bincount(r0[deg == k], minlength=k)
        n = cnt.sum()
        assert np.abs(cnt / n - 1.0 / k).max() < 5 * np.sqrt((1.0 / k) / n), (k, cnt)
