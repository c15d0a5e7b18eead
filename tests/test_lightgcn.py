"""LightGCN (BASELINE config 5): CPU oracle vs the reference goldens
(not gpu) and the HIP path vs both (gpu)."""
import os
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, golden, rel_err, split_batches

G6 = ["g6_lightgcn_50x40_d32_L1", "g6_lightgcn_50x40_d32_L2", "g6_lightgcn_50x40_d32_L3",
      "g6_lightgcn_ml100k_d64_L2", "g6_lightgcn_200x150_d128_L3",     # BASELINE configs[4] model shape
      "g6_lightgcn_120x90_d50_L4_isolated"]                             # (round 5) emb_dim 50 (stored as 64), isolated nodes, 4 layers


def graph_of(g):
    N = len(g["A_indptr"]) - 1
    return sp.csr_matrix((g["A_data"], g["A_indices"], g["A_indptr"]), shape=(N, N))


def ratings_of(g):
    U, I = g["P0"].shape[0], g["Q0"].shape[0]
    return sp.csr_matrix((np.ones(len(g["R_indices"]), np.float32), g["R_indices"], g["R_indptr"]), shape=(U, I))


@pytest.mark.parametrize("name", G6)
def test_oracle_lightgcn_matches_reference(oracle_mod, name):
    g = golden(name)
    A = oracle_mod.normalized_adjacency(ratings_of(g))          # LightGCN.py:228-258
    assert np.array_equal(A.indptr, g["A_indptr"]) and np.array_equal(A.indices, g["A_indices"])
    assert np.allclose(A.data, g["A_data"], rtol=2e-7, atol=0)
    m = oracle_mod.LightGCNOracle(g["P0"], g["Q0"], graph_of(g), int(g["num_layers"]))
    ou, oi = m.propagate()                                       # LightGCN.py:174-202
    assert rel_err(ou, g["out0_u"]) < 2e-6 and rel_err(oi, g["out0_i"]) < 2e-6
    for t, (u, i, j) in enumerate(split_batches(g)):
        assert abs(m.step(u, i, j) - g["loss"][t]) < 1e-5
    assert rel_err(m.P, g["PT"]) < 1e-5 and rel_err(m.Q, g["QT"]) < 1e-5


def test_product_adjacency_equals_the_oracle_one(oracle_mod):
    from recsys_pytorch_amd.lightgcn import normalized_adjacency
    g = golden(G6[3])
    A = normalized_adjacency(ratings_of(g))
    assert np.array_equal(A.indptr, g["A_indptr"]) and np.array_equal(A.indices, g["A_indices"])
    assert np.allclose(A.data, g["A_data"], rtol=2e-7, atol=0)


def test_spmm_plan_covers_every_nonzero_once():
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(0, 5, 50), [1000, 0, 129, 128, 127]])
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = len(lens)
    cnt = rsx.lib().rsx_spmm_plan(indptr.ctypes.data, n, 128, None, None, None)
    row, beg, ln = np.empty(cnt, np.int32), np.empty(cnt, np.int64), np.empty(cnt, np.int32)
    rsx.lib().rsx_spmm_plan(indptr.ctypes.data, n, 128, row.ctypes.data, beg.ctypes.data, ln.ctypes.data)
    assert ln.max() <= 128 and ln.sum() == lens.sum()
    assert sorted(set(row)) == list(range(n))                    # empty rows get a segment too
    for r in range(n):
        segs = np.nonzero(row == r)[0]
        assert beg[segs[0]] == indptr[r] and (beg[segs] + ln[segs])[-1] == indptr[r + 1]


@pytest.mark.gpu
@pytest.mark.parametrize("name", G6)
def test_hip_lightgcn_matches_reference_golden(name):
    import recsys_pytorch_amd as pkg
    g = golden(name)
    U, I, d = g["P0"].shape[0], g["Q0"].shape[0], g["P0"].shape[1]
    ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="t")
    m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": int(g["num_layers"]), "node_dropout": 0.0, "split": False,
                          "num_folds": 100, "reg": 1e-4, "graph_dir": "graph"}, "cuda")
    m.load_tables(g["P0"], g["Q0"])
    m.getSparseGraph(ratings_of(g))
    m.update_lightgcn_embedding()
    assert rel_err(m.user_embeddings[:, :d].cpu().numpy(), g["out0_u"]) < 2e-6
    assert rel_err(m.item_embeddings[:, :d].cpu().numpy(), g["out0_i"]) < 2e-6
    for t, (u, i, j) in enumerate(split_batches(g)):
        assert abs(float(m.train_step(u, i, j)) - g["loss"][t]) < 1e-5
    assert rel_err(m.user_embedding.weight.cpu().numpy(), g["PT"]) < 1e-5
    assert rel_err(m.item_embedding.weight.cpu().numpy(), g["QT"]) < 1e-5


@pytest.mark.gpu
def test_hip_lightgcn_evaluation_equals_the_reference(oracle_mod):
    """conf/LightGCN.yaml's shape on ml-100k: after the golden's six steps the reference evaluated its model (main.py:62-63, its Evaluator on the
    valid split; oracle/gen_golden_lightgcn.py) -- the package's model + Evaluator give the same top-10 wherever the 10th / 11th scores
    are apart, and the same score dictionary"""
    import recsys_pytorch_amd as pkg
    g, e, c = golden("g6_lightgcn_ml100k_d64_L2"), golden("g4_eval_lightgcn_ml100k"), golden("ml100k_csr")
    U, I, d = g["P0"].shape[0], g["Q0"].shape[0], g["P0"].shape[1]
    ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="t")
    m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": int(g["num_layers"]), "node_dropout": 0.0, "split": False, "num_folds": 100, "reg": 1e-4,
                          "graph_dir": "graph"}, "cuda")
    m.load_tables(g["P0"], g["Q0"])
    R = ratings_of(g)
    m.getSparseGraph(R)
    for u, i, j in split_batches(g):
        m.train_step(u, i, j)
    valid = sp.csr_matrix((np.ones(len(c["valid_indices"]), np.float32), c["valid_indices"].astype(np.int32), c["valid_indptr"]), shape=(U, I))
    top = m.predict_topk(np.arange(U), R, 10)
    # (six Adam steps at lr 1e-3 from the reference's small initial tables: the scores are of order 1e-3 and the 10th / 11th of a row
    #  often closer than two fp32 summation orders resolve -- "apart" is relative to the largest score)
    safe = e["gap_10"] > 2e-5 * float(e["score_max"])
    assert safe.mean() > 0.5 and all(set(top[r]) == set(e["topk10"][r]) for r in np.nonzero(safe)[0])
    scores = pkg.Evaluator(R, valid, "holdout", [5, 10]).evaluate(m)
    for n, want in zip(e["names"], e["scores_py"]):
        assert abs(float(scores[str(n)]) - float(want)) <= 1e-6 + (~safe).sum() / U * 0.2, (str(n), float(scores[str(n)]), float(want))


@pytest.mark.gpu
@pytest.mark.parametrize("max_seg", [None, 128, 7])
def test_hip_spmm_long_rows_vs_oracle(oracle_mod, max_seg):
    """rows far longer than a segment (split + atomics into rows the product clears itself), empty rows, d = 32/64/128; the
    default segment length (1024), the round-2 one and a tiny one"""
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(5)
    N = 3000
    A = sp.random(N, N, density=0.004, format="lil", random_state=rng, dtype=np.float32)
    A[7, :] = rng.random(N).astype(np.float32)                   # a 3000-neighbour row
    A[11, ::2] = 1.0
    A = A.tocsr()
    A[5, :] = 0
    A.eliminate_zeros()
    G = rsx.SpmmGraph(A, "cuda", max_seg=max_seg)
    seg = max_seg or 1024
    for d in (32, 64, 128, 256):
        X = rng.standard_normal((N, d)).astype(np.float32)
        Yo = np.empty_like(X)
        oracle_mod.lib().orc_spmm_csr(A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32),
                                      X, Yo, N, d)
        Xd = torch.from_numpy(X).cuda()
        Y = torch.full_like(Xd, 7.0)
        S = Xd.clone()
        rsx.spmm(G, Xd, Y, S_acc=S)
        # (3000-term fp32 sums in another order than the oracle's sequential one: 2.2e-6 of the largest entry on the d = 256 draw)
        assert rel_err(Y.cpu().numpy(), Yo) < 4e-6
        assert rel_err(S.cpu().numpy(), X + Yo) < 4e-6
        assert float(Y[5].abs().max()) == 0.0
        # an X most of whose rows are zero, with its row flags: the flagged-off rows are not fetched, the result is BIT-identical
        # (single-segment rows; split rows add their partial sums atomically, whose order is free: to rounding)
        Xs = Xd.clone()
        dead = torch.from_numpy(rng.random(N) < 0.9).cuda()
        Xs[dead] = 0.0
        flags = (~dead).to(torch.uint8)
        Ya, Yb = torch.empty_like(Xd), torch.empty_like(Xd)
        Sa, Sb = Xs.clone(), Xs.clone()
        rsx.spmm(G, Xs, Ya, S_acc=Sa)
        rsx.spmm(G, Xs, Yb, S_acc=Sb, x_nonzero=flags)
        short = torch.from_numpy(np.diff(A.indptr) <= seg).cuda()
        assert torch.equal(Ya[short], Yb[short]) and torch.equal(Sa[short], Sb[short])
        assert float((Ya - Yb).abs().max()) <= 2e-6 * float(Ya.abs().max())
        # flags that cover only part of the non-zero rows DROP the rest: they are the caller's claim
        part = flags.clone(); part[::2] = 0
        rsx.spmm(G, Xs, Yb, x_nonzero=part)
        Xp = Xs.clone(); Xp[part == 0] = 0.0
        rsx.spmm(G, Xp, Ya)
        assert torch.equal(Ya[short], Yb[short])
        # the first product of a propagation fused with the start of the running sum (rsx_spmm_csr_init): S = S_init + A X
        Si = torch.full_like(Xd, 123.0)
        Yi = torch.empty_like(Xd)
        rsx.spmm(G, Xd, Yi, S_acc=Si, S_init=Xd)
        assert torch.equal(Yi[short], Y[short]) and torch.equal(Si[short], S[short])     # S above = X.clone() then += A X
        assert float((Si - S).abs().max()) <= 4e-6 * float(S.abs().max())
        Zr = Xd.clone()
        fl = torch.from_numpy((np.random.default_rng(5 + d).random(N) < 0.3).astype(np.uint8)).cuda()
        rsx.scale_rows(Zr, fl, 0.25)
        assert torch.equal(Zr[fl.bool()], Xd[fl.bool()] * 0.25) and torch.equal(Zr[~fl.bool()], Xd[~fl.bool()])
        Zr[0, 0] = float("inf")
        fl[0] = 1
        rsx.scale_rows(Zr, fl, 0.0)                               # alpha = 0 CLEARS (whatever the row held)
        assert float(Zr[fl.bool()].abs().max()) == 0.0 and torch.equal(Zr[~fl.bool()], Xd[~fl.bool()])
        # only SOME rows of the result wanted (rsx_spmm_csr_select_rows: the last forward product of a LightGCN step): those rows
        # equal the full product's (bit for bit where a row is one segment), the others keep what they held -- in Y and in S_acc
        want = torch.from_numpy((np.random.default_rng(77 + d).random(N) < 0.2).astype(np.uint8)).cuda()   # (own generator: the draws above stay what they were)
        want[7] = 1; want[11] = 0                                # one of the long (split) rows wanted, the other not
        Yw, Sw = torch.full_like(Xd, 7.0), torch.full_like(Xd, -3.0)
        rsx.spmm(G, Xd, Yw, S_acc=Sw, y_wanted=want)
        w = want.bool()
        assert torch.equal(Yw[w & short], Y[w & short]) and float((Yw[w] - Y[w]).abs().max()) <= 2e-6 * float(Y.abs().max())
        assert float((Sw[w] - (Y[w] - 3.0)).abs().max()) <= 4e-6 * float(Y.abs().max() + 3.0)
        assert bool((Yw[~w] == 7.0).all()) and bool((Sw[~w] == -3.0).all())      # unwanted rows -- split ones too (row 11) -- are not written


@pytest.mark.gpu
def test_hip_spmm_on_random_graphs():
    """20 random square sparse matrices (empty rows, rows of one entry, heavy tails, rows holding every column; any segment length from 1
    up; every row width) through every form of the product -- plain, with the running sum, started from S_init, with row flags on X,
    with only some rows wanted -- against scipy in fp64; every other graph with its longest rows computed by scatter"""
    from conftest import fuzz
    from recsys_pytorch_amd import rsx
    rng, trials = fuzz(2024, 20)
    for trial in range(trials):
        N = int(rng.integers(1, 4000))
        d = int(rng.choice([32, 64, 128, 256]))
        kind = trial % 4
        degs = (rng.integers(0, min(N, 4) + 1, N) if kind == 0 else rng.integers(0, min(N, 60) + 1, N) if kind == 1
                else np.where(rng.random(N) < 0.01, N, rng.integers(0, min(N, 8) + 1, N)) if kind == 2
                else np.minimum(N, (rng.pareto(0.8, N) * 2).astype(np.int64)))
        cols = [np.sort(rng.choice(N, int(g), replace=False)) for g in degs]
        indptr = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
        indices = (np.concatenate(cols) if indptr[-1] else np.zeros(0)).astype(np.int32)
        vals = rng.standard_normal(len(indices)).astype(np.float32)
        A = sp.csr_matrix((vals, indices, indptr), shape=(N, N))
        max_seg = [None, 1, 3, 64, 1000][int(rng.integers(0, 5))]
        # every other graph with its longest rows by scatter (include/rsx.h: rsx_spmm_hot_rows), forced (these graphs are small)
        hot = bool(trial % 2)
        ctx = f"trial {trial}: N={N} d={d} kind={kind} nnz={len(indices)} max_seg={max_seg} hot={hot}"
        G = rsx.SpmmGraph(A, "cuda", max_seg=max_seg, d=d, hot=hot)
        assert (G.hot is not None) == (hot and len(indices) > 0), ctx
        # (found by the campaign, seed 6: a graph with no empty row and fewer non-empty rows than hot slots has NO segment left)
        X = rng.standard_normal((N, d)).astype(np.float32)
        A64 = A.astype(np.float64)
        Y64 = A64 @ X.astype(np.float64)
        # the bound: fp32 sums of up to N terms in some order -- a few ulp of the sum of the magnitudes
        bound = 4e-6 * np.maximum((abs(A64) @ np.abs(X).astype(np.float64)), 1e-30).max()
        Xd = torch.from_numpy(X).cuda()
        Y, S = torch.full_like(Xd, 7.0), Xd.clone()
        rsx.spmm(G, Xd, Y, S_acc=S)
        assert np.abs(Y.cpu().numpy() - Y64).max() <= bound, ctx
        assert np.abs(S.cpu().numpy() - (X + Y64)).max() <= bound + 1e-6 * np.abs(X).max(), ctx
        empty = torch.from_numpy(np.diff(indptr) == 0).cuda()
        assert not bool(empty.any()) or float(Y[empty].abs().max()) == 0.0, ctx
        Yi, Si = torch.full_like(Xd, -1.0), torch.full_like(Xd, 123.0)
        rsx.spmm(G, Xd, Yi, S_acc=Si, S_init=Xd)
        assert np.abs(Yi.cpu().numpy() - Y64).max() <= bound and np.abs(Si.cpu().numpy() - (X + Y64)).max() <= bound + 1e-6 * np.abs(X).max(), ctx
        dead = rng.random(N) < 0.7
        Xs = X.copy(); Xs[dead] = 0.0
        Ys64 = A64 @ Xs.astype(np.float64)
        Yb = torch.full_like(Xd, 5.0)
        rsx.spmm(G, torch.from_numpy(Xs).cuda(), Yb, x_nonzero=torch.from_numpy((~dead).astype(np.uint8)).cuda())
        assert np.abs(Yb.cpu().numpy() - Ys64).max() <= bound, ctx
        want = rng.random(N) < 0.3
        Yw, Sw = torch.full_like(Xd, 7.0), torch.full_like(Xd, -3.0)
        rsx.spmm(G, Xd, Yw, S_acc=Sw, y_wanted=torch.from_numpy(want.astype(np.uint8)).cuda())
        Yw, Sw = Yw.cpu().numpy(), Sw.cpu().numpy()
        assert not want.any() or (np.abs(Yw[want] - Y64[want]).max() <= bound and np.abs(Sw[want] - (Y64[want] - 3.0)).max() <= bound + 3e-6), ctx
        assert want.all() or (bool((Yw[~want] == 7.0).all()) and bool((Sw[~want] == -3.0).all())), ctx


@pytest.mark.gpu
@pytest.mark.parametrize("N,d", [(50, 128), (120, 32), (7, 256)])
def test_hip_spmm_when_every_row_goes_by_scatter(N, d):
    """a graph with no empty row and fewer rows than hot slots: the segment plan is left EMPTY and the whole product is the scatter
    (found by the random-shape campaign, seed 6: the planned product was called with no segment)"""
    from recsys_pytorch_amd import rsx
    rng = np.random.default_rng(N + d)
    A = sp.random(N, N, density=0.3, format="csr", random_state=rng, dtype=np.float32)
    A = (A + sp.identity(N, dtype=np.float32, format="csr")).tocsr()          # every row owns an entry
    G = rsx.SpmmGraph(A, "cuda", d=d, hot=True)
    assert G.hot is not None and G.num_segs == 0
    X = rng.standard_normal((N, d)).astype(np.float32)
    Xd = torch.from_numpy(X).cuda()
    Y, S = torch.full_like(Xd, 7.0), Xd.clone()
    rsx.spmm(G, Xd, Y, S_acc=S)
    Y64 = A.astype(np.float64) @ X.astype(np.float64)
    bound = 4e-6 * np.maximum(abs(A.astype(np.float64)) @ np.abs(X).astype(np.float64), 1e-30).max()
    assert np.abs(Y.cpu().numpy() - Y64).max() <= bound and np.abs(S.cpu().numpy() - (X + Y64)).max() <= bound + 1e-6 * np.abs(X).max()


@pytest.mark.gpu
def test_hip_lightgcn_on_random_graphs(oracle_mod):
    """10 random interaction matrices (users and items WITHOUT any interaction -- isolated nodes, zero rows of the adjacency -- short
    and long rows; any emb_dim in 1 .. 256; 1-4 layers) through the model class: the propagated embeddings (LightGCN.py:174-202)
    against the oracle's to 2e-6, the loss of the first two steps to 1e-5, and after them the base tables within two Adam steps
    of the oracle's (|delta| <= 2 lr everywhere and the mean far below: an Adam step is +-lr at any non-zero gradient entry,
    so entries whose gradient nearly cancels may differ by a step -- the goldens G6 pin whole trajectories)"""
    import recsys_pytorch_amd as pkg
    from conftest import fuzz
    rng, trials = fuzz(606, 10)
    for trial in range(trials):
        U, I = int(rng.integers(2, 900)), int(rng.integers(2, 700))
        d = int(rng.integers(1, 257)) if trial % 2 else int(rng.choice([32, 64, 128, 256]))
        L = int(rng.integers(1, 5))
        dens = float(rng.choice([0.002, 0.02, 0.2]))
        R = sp.random(U, I, density=dens, format="csr", random_state=np.random.default_rng(1000 + trial), dtype=np.float32)
        R.data[:] = 1.0
        if R.nnz == 0:
            R = sp.csr_matrix(([1.0], ([0], [0])), shape=(U, I), dtype=np.float32)
        B = int(rng.integers(1, 600))
        ctx = f"trial {trial}: U={U} I={I} emb_dim={d} L={L} nnz={R.nnz} B={B}"
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="t")
        m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 100, "reg": 1e-4,
                              "graph_dir": "graph"}, "cuda")
        m.load_tables(P0, Q0)
        m.getSparseGraph(R)
        orc = oracle_mod.LightGCNOracle(P0, Q0, oracle_mod.normalized_adjacency(R), L)
        m.update_lightgcn_embedding()
        ou, oi = orc.propagate()
        scale = max(np.abs(ou).max(), np.abs(oi).max(), 1e-30)
        assert np.abs(m.user_embeddings[:, :d].cpu().numpy() - ou).max() <= 2e-6 * scale, ctx
        assert np.abs(m.item_embeddings[:, :d].cpu().numpy() - oi).max() <= 2e-6 * scale, ctx
        for t in range(2):
            u, i, j = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
            want = orc.step(u, i, j)
            got = float(m.train_step(u, i, j))
            assert abs(got - want) < 1e-5 * max(1.0, abs(want)), (ctx, t)
        E = np.concatenate([m.user_embedding.weight.cpu().numpy(), m.item_embedding.weight.cpu().numpy()])
        diff = np.abs(E - orc.E0)
        assert diff.max() <= 2.0 * orc.lr * 1.001 + 1e-7 and diff.mean() <= 0.02 * orc.lr, (ctx, float(diff.max()), float(diff.mean()))


@pytest.mark.gpu
def test_hip_lightgcn_fit_and_topk_end_to_end():
    """fit() on ml-100k with the evaluator: loss goes down, predict and predict_topk agree"""
    import recsys_pytorch_amd as pkg
    ds = pkg.InteractionData.from_npz(os.path.join(GOLDEN, "ml100k_csr.npz"))
    ds.dataname = "ml-100k"
    torch.manual_seed(1)
    m = pkg.LightGCN(ds, {"emb_dim": 64, "num_layers": 2, "node_dropout": 0.0, "split": False, "num_folds": 100,
                          "reg": 1e-4, "graph_dir": "graph", "lr": 5e-3}, "cuda")
    ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "holdout", [10])
    logged = []
    cfg = types.SimpleNamespace(batch_size=256, num_epochs=12, verbose=0, test_from=12, test_step=1)
    m.getSparseGraph(ds.train_data)                              # (the reference builds it in fit, :70)
    before = ev.evaluate(m)
    ret = m.fit(ds, cfg, evaluator=ev, loggers=[types.SimpleNamespace(log_metrics=lambda d, epoch: logged.append(d))])
    assert logged[-1]["loss"] < logged[0]["loss"]
    assert ret["scores"]["NDCG@10"] > before["NDCG@10"]
    users = np.arange(64)
    pred = m.predict(users, ds.train_data, 32)[:64]
    top = m.predict_topk(users, ds.train_data, 10)
    for r in range(64):
        kth = np.sort(pred[r])[::-1][9]
        assert np.all(pred[r][top[r]] >= kth) and np.all(np.isfinite(pred[r][top[r]]))
    # a further fit on the SAME matrix keeps the graph (models/LightGCN.py:70 builds it in every fit; 1.9 s of host work at 1M x 100K) and
    # the device copy of the matrix; another matrix builds both again
    built = []
    real = m.getSparseGraph
    m.getSparseGraph = lambda R, adjacency=None: (built.append(R.shape), real(R, adjacency))[1]
    one = types.SimpleNamespace(batch_size=256, num_epochs=1, verbose=0, test_from=1, test_step=1)
    g0 = m.Graph
    m.fit(ds, one); m.fit(ds, one)
    assert built == [] and m.Graph is g0
    import scipy.sparse as sp
    other = sp.csr_matrix(ds.train_data, copy=True)
    other.data[:] = 1.0
    other[0, :5] = 1.0
    ds2 = pkg.InteractionData(sp.csr_matrix(other), ds.valid_target, ds.test_target)
    m.fit(ds2, one)
    assert built == [ds.train_data.shape] and m.Graph is not g0


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_hip_lightgcn_full_size_config5_properties():
    """BASELINE configs[4] at its full shape: 1M users x 100K items, d=128, 3 layers (nnz(A_hat) =
    4e7), through properties that hold at any size:
      * sqrt(degree) is an eigenvector of A_hat = D^-1/2 A D^-1/2 with eigenvalue 1, so tables whose
        columns all equal sqrt(deg) propagate to themselves (every layer, hence their mean);
      * A_hat is symmetric: <A x, y> == <x, A y>;
      * sampled output rows equal an fp64 gather over the CSR row;
      * one training step (LightGCN.py:83-87): loss = mean softplus(-x) on the propagated tables,
        Adam's first step moves a parameter by lr |g| / (|g| + eps): never more than lr."""
    import recsys_pytorch_amd as pkg
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    U, I, d, L, deg = 1_000_000, 100_000, 128, 3, 20
    ip, ix = synthetic_csr(U, I, deg, "cuda", seed=2020)
    R = sp.csr_matrix((np.ones(U * deg, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
    ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="c5")
    torch.manual_seed(5)
    m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 100,
                          "reg": 1e-4, "graph_dir": "graph"}, "cuda")
    g = m.getSparseGraph(R)
    N = U + I
    assert g.n == N and int(g.indices.numel()) == 2 * U * deg
    # eigenvector
    degs = torch.cat([torch.full((U,), float(deg), device="cuda"), torch.bincount(ix.long(), minlength=I).float()])
    E0_saved = m._E0.clone()
    m._E0.copy_(degs.sqrt().unsqueeze(1).expand(N, d))
    m.update_lightgcn_embedding()
    # (the most popular item row sums ~7e5 equal positive terms in fp32: equal addends round the same
    #  way, so the error grows linearly, ~2e-4 relative there -- in any summation order, the reference's
    #  sequential torch.sparse.mm included)
    assert float(((m._out - m._E0).abs() / m._E0.abs()).max()) < 1e-3
    assert float(((m._out - m._E0).abs() / m._E0.abs())[:U].max()) < 1e-4          # user rows: 20 neighbours, fed by those item rows
    # symmetry on random vectors (fp64 inner products)
    x, y = torch.randn(N, d, device="cuda"), torch.randn(N, d, device="cuda")
    ax, ay = torch.empty_like(x), torch.empty_like(y)
    rsx.spmm(g, x, ax); rsx.spmm(g, y, ay)
    lhs, rhs = float((ax.double() * y.double()).sum()), float((x.double() * ay.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), abs(rhs), float(ax.double().norm() * y.double().norm()) * 1e-3)
    # sampled rows against an fp64 gather (users: 20 neighbours; items: up to 1e5+)
    ipn = g.indptr.cpu().numpy()
    for r in [0, 1, U - 1, U, U + 1, U + 5, U + 999, N - 1] + list(np.random.default_rng(0).integers(0, N, 24)):
        lo, hi = int(ipn[r]), int(ipn[r + 1])
        want = (g.vals[lo:hi].double().unsqueeze(1) * x[g.indices[lo:hi].long()].double()).sum(0)
        assert float((ax[r].double() - want).abs().max()) < 1e-5 * max(float(want.abs().max()), 1e-3), r
    del x, y, ax, ay
    # one training step at the reference's init (N(0, 0.01), LightGCN.py:50-51)
    m._E0.copy_(E0_saved)
    B = 65_536
    rng = np.random.default_rng(7)
    u, i, j = rng.permutation(U)[:B], rng.integers(0, I, B), rng.integers(0, I, B)
    m.update_lightgcn_embedding()
    ut, it, jt = (torch.from_numpy(a).cuda().long() for a in (u, i, j))
    xs = (m._out[:U][ut].double() * (m._out[U:][it].double() - m._out[U:][jt].double())).sum(1)
    want_loss = float(torch.nn.functional.softplus(-xs).mean())
    loss = float(m.train_step(u, i, j))
    assert abs(loss - want_loss) < 1e-5
    step = (m._E0 - E0_saved).abs()
    assert bool(torch.isfinite(m._E0).all())
    # Adam's first step is lr * g / (|g| + eps): at most lr, and close to lr where |g| >> eps = 1e-8
    assert float(step.max()) <= 1e-3 * (1 + 1e-3) and float(step.max()) > 0.5e-3
    assert int((step > 0).any(1).sum()) > B                      # propagation spreads the gradient beyond the batch


@pytest.mark.gpu
def test_mark_batch_rows_flags_exactly_the_rows_of_the_batch():
    """include/rsx.h:rsx_spmm_mark_batch_rows -- the row flags of the first backward product (LightGCN.train_step): 1 at the
    batch's users and (offset by U) at its positive and negative items, 0 elsewhere; skipped triplets (i < 0) mark nothing;
    stale flags of the batch before are cleared"""
    import torch
    from recsys_pytorch_amd import rsx
    U, I, B = 5000, 1200, 700
    g = torch.Generator().manual_seed(5)
    flags = torch.ones(U + I, dtype=torch.uint8, device="cuda")             # (stale content)
    u = torch.randint(0, U, (B,), generator=g).int()
    i = torch.randint(0, I, (B,), generator=g).int()
    j = torch.randint(0, I, (B,), generator=g).int()
    i[::9] = -1; j[::9] = -1
    rsx.mark_batch_rows(flags, u.cuda(), i.cuda(), j.cuda(), U)
    want = torch.zeros(U + I, dtype=torch.uint8)
    live = i >= 0
    want[u[live].long()] = 1; want[U + i[live].long()] = 1; want[U + j[live].long()] = 1
    assert torch.equal(flags.cpu(), want)
    rsx.mark_batch_rows(flags, u[:0].cuda(), i[:0].cuda(), j[:0].cuda(), U)  # an empty batch clears everything
    assert int(flags.sum()) == 0
