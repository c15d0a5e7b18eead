"""GPU: the step as a pipeline over item ranges (include/rsx.h: "item chunks") -- the chunked sampler's rules, the chunked
step kernel against the CPU oracle on the dumped triplets, the range-by-range apply on the trainer's own stream, the
contract check, and the full-size step through the native loop.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest
import torch

from conftest import UPDATE_TOL, assert_update, fuzz, resolvable_lr

pytestmark = pytest.mark.gpu


def _engine(U, I, d, B, deg, chunks, lr, seed=3, hot=0, max_block=8, pop="zipf"):
    """B >= 2 I: the blocked layout (negatives from an item block of the positive's range); below: the ranges WITHOUT blocks
    (include/rsx.h "item chunks", neg_block = 0: negatives uniform over the real items of the positive's range)"""
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    ip, ix = synthetic_csr(U, I, deg, "cuda", seed=seed, popularity=pop)
    torch.manual_seed(seed)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(I, d, device="cuda") * 0.1
    eng = BPREngine(P, Q, lr)
    nb = eng.set_neg_block(B, max_block)
    assert (nb > 0) == (B >= 2 * I)
    if nb == 0:
        eng.sorted_min_batch = 1          # (test sizes: the ordered layout engages from min(2 I, 2^19) triplets on by default)
    if hot:
        eng.set_hot_items(torch.bincount(ix.long(), minlength=I), hot, 4)
    eng.set_chunks(chunks)
    return eng, P, Q, ip, ix


@pytest.mark.parametrize("U,I,d,B,deg,chunks,hot", [(30_000, 5_000, 128, 30_000, 12, 4, 32), (20_000, 3_001, 64, 9_000, 8, 3, 0),
                                                     (50_000, 7_777, 32, 50_000, 10, 8, 16), (12_000, 1_000, 128, 12_000, 6, 2, 8),
                                                     # B < 2 I: the ranges without blocks (the configs[3] shape in small)
                                                     (30_000, 40_000, 128, 30_000, 12, 2, 32), (20_000, 15_001, 64, 20_000, 8, 4, 0),
                                                     (25_000, 30_011, 32, 11_000, 10, 3, 16)])
def test_chunked_native_steps_follow_the_range_rule_and_replay_through_the_oracle(oracle_mod, U, I, d, B, deg, chunks, hot):
    """every native chunked step: users unique, batch ordered by (relabelled) positive item, positions of range k hold
    positives AND negatives of range k only (real items, never padding rows), true positives / negatives; the dumped
    triplets replayed on the CPU oracle (same relabelled tables) give the same loss and -- with a resolvable step size --
    the same UPDATE of P and Q to 1e-5 of its size; adopt() brings the item rows back to the caller's ids"""
    from recsys_pytorch_amd import rsx
    lr = resolvable_lr(B)
    eng, P, Q, ip, ix = _engine(U, I, d, B, deg, chunks, lr, hot=hot)
    Q_nat0 = Q.clone()
    acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
    tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
    assert tr.chunks == chunks
    r = eng._relabel
    Ic, nb = r["Ic"], eng.neg_block
    assert (nb > 0) == (B >= 2 * I)
    assert Ic % max(nb, 1) == 0 and Ic * chunks == r["Q"].shape[0] >= I and (nb > 0 or Ic == -(-I // chunks))
    # the relabelling is a bijection onto the real rows, and the relabelled table holds the same item rows
    rank_item = r["rank_item"].cpu().numpy()
    real = rank_item >= 0
    assert real.sum() == I and len(np.unique(rank_item[real])) == I
    assert torch.equal(r["Q"][r["real"]], Q_nat0[r["rank_item"][r["real"]]]) and float(r["Q"][~r["real"]].abs().max() if (~real).any() else 0) == 0.0
    base, rem = divmod(I, chunks)
    n_real = np.array([base + (k < rem) for k in range(chunks)])
    assert all(real[k * Ic:k * Ic + n_real[k]].all() and not real[k * Ic + n_real[k]:(k + 1) * Ic].any() for k in range(chunks))
    P0, Qm0 = P.cpu().numpy(), r["Q"].cpu().numpy()
    orc = oracle_mod.MFOracle(P0, Qm0, "sgd", lr)
    ipn, ixm = ip.cpu().numpy(), r["indices"].cpu().numpy()
    neg_hist = np.zeros(chunks * Ic)
    for step in range(3):
        acc.zero_()
        tr.run(1)
        torch.cuda.synchronize()
        u, i, j, nb_ran, key = tr.last_batch()
        cp = tr.last_chunk_pos().cpu().numpy()
        un, inn, jn = u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()
        live = inn >= 0
        n_live = int(live.sum())
        assert nb_ran == nb and (key != 0) == (nb > 0) and len(np.unique(un)) == B
        assert live[:n_live].all() and (jn[~live] == -1).all()                       # skipped pairs come last
        assert cp[0] == 0 and cp[-1] == n_live and np.all(np.diff(cp) >= 0)
        assert np.all(np.diff(inn[:n_live]) >= 0)                                    # ordered by (relabelled) positive item
        for k in range(chunks):
            sl = slice(cp[k], cp[k + 1])
            assert np.all(inn[sl] // Ic == k), k                                      # positives of range k ...
            assert np.all(jn[sl] // Ic == k), k                                       # ... and their negatives: range k only
            assert np.all(jn[sl] - k * Ic < n_real[k]) and np.all(inn[sl] - k * Ic < n_real[k])    # real rows, never padding
        for a, b_, c_ in list(zip(un, inn, jn))[:n_live:max(1, B // 400)]:
            row = ixm[ipn[a]:ipn[a + 1]]
            assert b_ in row and c_ not in row                                       # true positive, true negative
        neg_hist += np.bincount(jn[live], minlength=chunks * Ic)
        assert n_live == B                                                           # (every user of these CSRs has a usable row)
        assert abs(float(acc.sum()) / B - orc.step(un, inn, jn)) < 1e-5
    tr.check()                                                                       # no triplet left its range, no wait timed out
    assert_update(P.cpu().numpy(), P0, orc.P, "P")
    assert_update(r["Q"].cpu().numpy(), Qm0, orc.Q, "Q (relabelled)")
    assert neg_hist[~real].sum() == 0 and neg_hist[real].min() >= 0
    exp = 3 * B / I
    assert abs(neg_hist[real].mean() - exp) < 0.02 * exp + 1e-9
    if nb == 0:      # no blocks: a position's negative is UNIFORM over the real items of its range -- every item of a range is
        for k in range(chunks):   # drawn at the same rate (the range's positions / its real items), up to counting noise
            h = neg_hist[k * Ic:k * Ic + n_real[k]]
            assert abs(h.std() - np.sqrt(h.mean())) < 0.15 * np.sqrt(h.mean()) + 0.05, (k, h.mean(), h.std())
    eng.adopt(tr)                                                                    # item rows back in the caller's ids
    assert torch.equal(Q[r["rank_item"][r["real"]]], r["Q"][r["real"]]) and not torch.equal(Q, Q_nat0)
    assert float(r["G"].abs().max()) == 0.0
    if r["hot"] is not None:
        assert float(r["hot"].ghot.abs().max()) == 0.0
    tr.close()


def test_native_loop_on_random_shapes(oracle_mod):
    """16 random (users, items, d, batch, popularity, row lengths, item ranges or none, replicated rows or none, ordered layout or
    not) engines through the native loop, three steps each: what every step consumed (rsx_bpr_trainer_last_batch) is a batch of unique
    users with true positives and true negatives, and replayed on the CPU oracle it gives the same loss and the same update"""
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.sharded import BPREngine
    rng, trials = fuzz(555, 16)
    failures = []
    for trial in range(trials):
        chunks = int(rng.choice([0, 0, 2, 3, 5]))
        d = int(rng.choice([32, 64, 128, 256]))
        I = int(rng.integers(16 * max(chunks, 1), 5000))
        U = int(rng.integers(50, 8000))
        B = U if trial % 3 == 0 else int(rng.integers(max(1, U // 8), U + 1))
        maxdeg = max(1, min(12, I // (2 * max(chunks, 1)) - 1))          # (no user owns half a range: every pair finds a negative)
        pop = 1.0 / (1.0 + np.arange(I)) ** float(rng.choice([0.0, 0.7, 1.0]))
        draws = rng.choice(I, size=(U, maxdeg), p=pop / pop.sum())
        keep = rng.random((U, maxdeg)) < 0.6
        keep[:, 0] = True
        rows = [np.unique(draws[uu][keep[uu]]) for uu in range(U)]
        indptr = np.concatenate([[0], np.cumsum([len(r_) for r_ in rows])]).astype(np.int64)
        indices = np.concatenate(rows).astype(np.int32)
        ip, ix = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        P, Q = torch.from_numpy(P0).cuda(), torch.from_numpy(Q0).cuda()
        lr = resolvable_lr(B)
        eng = BPREngine(P, Q, lr, seed=int(rng.integers(1, 1 << 30)))
        nb = eng.set_neg_block(B, int(rng.integers(1, 17)))
        ordered = nb > 0
        if nb == 0 and (chunks > 1 or trial % 2):
            eng.sorted_min_batch = 1                                     # the ordered layout without blocks
            ordered = True
        hot = int(rng.choice([0, 8, 32]))
        if hot:
            eng.set_hot_items(torch.bincount(ix.long(), minlength=I), min(hot, I), int(rng.choice([1, 4])))
        if chunks > 1:
            eng.set_chunks(chunks)
        acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
        tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
        ranged = tr.chunks > 1
        ctx = f"trial {trial}: U={U} I={I} d={d} B={B} chunks={chunks} (ran {tr.chunks}) nb={nb} ordered={ordered} hot={hot} maxdeg={maxdeg}"
        try:
            assert ranged == (chunks > 1), ctx
            if ranged:
                r = eng._relabel
                Pn, Qn, ixn = P0, r["Q"].cpu().numpy(), r["indices"].cpu().numpy()
                table = r["Q"]
            else:
                Pn, Qn, ixn, table = P0, Q0, indices, Q
            orc = oracle_mod.MFOracle(Pn, Qn, "sgd", lr)
            for step in range(3):
                acc.zero_()
                tr.run(1)
                torch.cuda.synchronize()
                u, i, j = (x.cpu().numpy() for x in tr.last_batch()[:3])
                assert len(np.unique(u)) == B and i.min() >= 0 and j.min() >= 0, ctx
                assert not ordered or np.all(np.diff(i) >= 0), ctx
                for a, b_, c_ in list(zip(u, i, j))[::max(1, B // 200)]:
                    row = ixn[indptr[a]:indptr[a + 1]]
                    assert b_ in row and c_ not in row, ctx
                want = orc.step(u, i, j)
                assert abs(float(acc.sum()) / B - want) < 1e-5 * max(1.0, abs(want)), ctx
            tr.check()
            assert_update(P.cpu().numpy(), P0, orc.P, "P, " + ctx)
            assert_update(table.cpu().numpy(), Qn, orc.Q, "Q, " + ctx)
            eng.adopt(tr)
            assert float(eng.G.abs().max()) == 0.0 or ranged, ctx
        except AssertionError as e:
            failures.append(str(e).splitlines()[0])
        finally:
            tr.close()
    assert not failures, "\n".join(failures)


_REPRO_CHILD = """
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r)
from recsys_pytorch_amd.sharded import BPREngine
from recsys_pytorch_amd.data import synthetic_csr
h = lambda t: hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:8]
out = []
for name, U, I, B, chunks, ordered, hot in (("blocked", 2400, 400, 1000, 0, False, 16), ("ordered", 2400, 782, 1000, 0, True, 0),
                                            ("plain", 2400, 782, 1000, 0, False, 8), ("ranges+blocks", 2400, 300, 1000, 3, False, 16),
                                            ("ranges", 2400, 782, 1000, 3, True, 0)):
    ip, ix = synthetic_csr(U, I, 7, "cuda", seed=3)
    torch.manual_seed(1)
    P, Q = torch.randn(U, 64, device="cuda") * 0.1, torch.randn(I, 64, device="cuda") * 0.1
    eng = BPREngine(P, Q, 0.05 * B, seed=11)
    eng.set_neg_block(B, 8)
    if ordered:
        eng.sorted_min_batch = 1
    if hot:
        eng.set_hot_items(torch.bincount(ix.long(), minlength=I), hot, 4)
    if chunks:
        eng.set_chunks(chunks)
    tr = eng.native_trainer(ip, ix, B)
    hs = [h(eng._relabel["rank_item"])] if chunks else []
    for _ in range(2):
        tr.run(1); torch.cuda.synchronize()
        hs += [h(x) for x in tr.last_batch()[:3]]
    out.append(name + ":" + "".join(hs))
    tr.close()
print("HASH", " ".join(out))
"""


def test_item_ranges_of_a_seeded_engine_are_the_same_in_every_process():
    """what a seeded engine samples is a function of the seed and the data in every layout (blocked, ordered, plain, item ranges with
    and without blocks): three PROCESSES deal the items to the ranges alike and sample the same first two batches.  (Until round 5 the items' sampling masses were float64 sums added atomically on the device: their last bits,
    and with them the dealing of items that tie, changed from launch to launch -- found by the random-shape test of
    tests/test_sharded_gloo.py, whose two arms are two launches; the masses are fixed-point integers now)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", _REPRO_CHILD % root], capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("HASH")]
        assert r.returncode == 0 and lines, r.stderr[-2000:]
        outs.append(lines[0])
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.parametrize("nb", [6, 0])
def test_chunked_step_counts_triplets_outside_their_range(nb):
    """the contract check: the same kernel on triplets that do NOT follow the range rule still sums them, but counts every
    one that touched a row outside its range (the native loop turns a non-zero count into an error); nb = 0: the form without
    blocks"""
    from recsys_pytorch_amd import rsx
    U, I, d, B, C = 8000, 1200, 64, 6000, 4
    Ic = rsx.chunk_rows(I, C, nb)
    rng = np.random.default_rng(0)
    P = torch.randn(U, d, device="cuda") * 0.1
    Q = torch.randn(C * Ic, d, device="cuda") * 0.1
    G = torch.zeros_like(Q)
    to = lambda a: torch.from_numpy(a.astype(np.int32)).cuda()
    i = np.sort(rng.integers(0, C * Ic, B))
    cp = torch.from_numpy(np.searchsorted(i, np.arange(C + 1) * Ic).astype(np.int64)).cuda()
    progress = torch.zeros(rsx.RSX_PROGRESS_WORDS, dtype=torch.int32, device="cuda")
    # honest triplets: negative in the positive's range
    j_ok = (i // Ic) * Ic + rng.integers(0, Ic, B)
    rsx.bpr_step_chunked(P, Q, G, I, C, to(rng.permutation(U)[:B]), to(i), to(j_ok), 0.1, 1.0 / B, cp, progress, nb, 77)
    torch.cuda.synchronize()
    pr = progress.cpu().numpy()
    assert pr[rsx.RSX_PROGRESS_VIOLATIONS] == 0
    # foreign triplets: negatives anywhere
    progress.zero_()
    j_bad = rng.integers(0, C * Ic, B)
    rsx.bpr_step_chunked(P, Q, G, I, C, to(rng.permutation(U)[:B]), to(i), to(j_bad), 0.1, 1.0 / B, cp, progress, nb, 77)
    torch.cuda.synchronize()
    n_bad = int((j_bad // Ic != i // Ic).sum())
    assert n_bad > B // 2 and progress.cpu().numpy()[rsx.RSX_PROGRESS_VIOLATIONS] == n_bad


def test_chunked_and_unchunked_native_loops_train_alike():
    """same model, same number of steps: the chunked pipeline (relabelled ranges) and the plain blocked layout reach the
    same BPR loss within noise -- and the chunked run really ran chunked"""
    from recsys_pytorch_amd import rsx
    U, I, d, B = 40_000, 4_000, 64, 40_000
    out = {}
    for chunks in (0, 4):
        eng, P, Q, ip, ix = _engine(U, I, d, B, 10, chunks, resolvable_lr(B) * 0.2, seed=5, hot=16)
        acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
        tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
        assert (tr.chunks > 1) == (chunks > 1)
        tr.run(60)
        acc.zero_()
        tr.run(5)
        torch.cuda.synchronize()
        eng.adopt(tr)
        out[chunks] = float(acc.sum()) / (5 * B)
        tr.close()
    assert out[0] < 0.685 and out[4] < 0.685 and abs(out[0] - out[4]) < 0.2 * (0.6931 - out[0]), out   # both learned (ln 2 = untrained), equally well


@pytest.mark.parametrize("chunks", [2, 4])
def test_full_size_chunked_step_through_the_native_loop(chunks):
    """the headline shape (1M users x 100K items, d = 128, B = 1M) through the chunked native loop, verified in fp64 on the
    device from rsx_bpr_trainer_last_batch on the relabelled tables: P rows, Q after the range-by-range apply, the loss"""
    from stepcheck import verify_step
    from recsys_pytorch_amd import rsx
    U, I, d, B = 1_000_000, 100_000, 128, 1_000_000
    lr = resolvable_lr(B)
    eng, P, Q, ip, ix = _engine(U, I, d, B, 20, chunks, lr, seed=2020, hot=256)
    loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
    tr = eng.native_trainer(ip, ix, B, loss_acc=loss)
    r = eng._relabel
    P0, Qm0 = P.clone(), r["Q"].clone()
    tr.run(1)
    torch.cuda.synchronize()
    tr.check()
    u, i, j, nb, key = tr.last_batch()
    assert int(i.min()) >= 0 and int(torch.bincount(u.long(), minlength=U).max()) == 1
    cp = tr.last_chunk_pos()
    Ic = r["Ic"]
    assert bool(((i.long() // Ic) == (j.long() // Ic)).all()) and int(cp[-1]) == B
    v = verify_step(P0, Qm0, P, r["Q"], u, i, j, lr, 1.0 / B)
    ctx = {k: (f"{x:.3e}" if isinstance(x, float) else x) for k, x in v.items()}
    assert abs(float(loss.double().sum()) / B - v["loss"]) < 1e-5, ctx
    assert v["max_dP"] > 1e-3 and v["max_dQ"] > 1e-3, ctx
    assert v["err_P"] <= UPDATE_TOL and v["err_Q"] <= UPDATE_TOL and v["untouched_rows_equal"], ctx
    assert float(r["G"].abs().max()) == 0.0
    tr.close()


@pytest.mark.parametrize("chunks", [2, 4])
def test_full_size_config3_slice_as_item_ranges_through_the_native_loop(chunks):
    """BASELINE configs[3] as one rank sees it (1.25M users x 1M items, d = 128, B = 1.25M < 2 I) through the chunked native
    loop WITHOUT blocks: the range rule on every triplet, then P rows, Q after the range-by-range apply and the loss verified in
    fp64 on the device from rsx_bpr_trainer_last_batch on the relabelled tables"""
    from stepcheck import verify_step
    from recsys_pytorch_amd import rsx
    U, I, d, B = 1_250_000, 1_000_000, 128, 1_250_000
    lr = resolvable_lr(B)
    eng, P, Q, ip, ix = _engine(U, I, d, B, 10, chunks, lr, seed=2020, hot=256)
    assert eng.neg_block == 0
    loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
    tr = eng.native_trainer(ip, ix, B, loss_acc=loss)
    assert tr.chunks == chunks
    r = eng._relabel
    P0, Qm0 = P.clone(), r["Q"].clone()
    tr.run(1)
    torch.cuda.synchronize()
    tr.check()
    u, i, j, nb, key = tr.last_batch()
    assert nb == 0 and int(i.min()) >= 0 and int(torch.bincount(u.long(), minlength=U).max()) == 1
    cp = tr.last_chunk_pos()
    Ic = r["Ic"]
    assert bool(((i.long() // Ic) == (j.long() // Ic)).all()) and int(cp[-1]) == B and bool((i[1:] >= i[:-1]).all())
    assert bool(r["real"][j.long()].all()) and bool(r["real"][i.long()].all())
    share = (cp[1:] - cp[:-1]).double() / B
    assert float((share - 1.0 / chunks).abs().max()) < 0.01, share          # the ranges carry the same share of the batch
    v = verify_step(P0, Qm0, P, r["Q"], u, i, j, lr, 1.0 / B)
    ctx = {k: (f"{x:.3e}" if isinstance(x, float) else x) for k, x in v.items()}
    assert abs(float(loss.double().sum()) / B - v["loss"]) < 1e-5, ctx
    assert v["max_dP"] > 1e-3 and v["max_dQ"] > 1e-3, ctx
    assert v["err_P"] <= UPDATE_TOL and v["err_Q"] <= UPDATE_TOL and v["untouched_rows_equal"], ctx
    assert float(r["G"].abs().max()) == 0.0
    tr.run(3)                                                                # the ranges of neighbouring steps interleave
    torch.cuda.synchronize()
    eng.adopt(tr)                                                            # checks the run
    assert bool(torch.isfinite(P).all()) and bool(torch.isfinite(eng.Q).all())
    tr.close()


def test_chunked_sampler_on_random_shapes():
    """30 random CSRs (empty rows, rows owning a whole item range or everything, heavy tails, tiny and ragged sizes, any block
    size, 2..8 ranges) straight through rsx_bpr_sample_chunked: users unique, live pairs first and ordered by item, first
    positions of the ranges consistent, positives true, negatives true / real / in the positive's range, a user owning its whole
    range skipped, twice the same call gives the same bits"""
    from recsys_pytorch_amd import rsx
    rng, trials = fuzz(321, 30)
    for trial in range(trials):
        C = int(rng.integers(2, 9))
        c = int(rng.integers(0, 17)) if trial % 5 else 0        # 0: no blocks -- negatives over the real items of the positive's range
        I = int(rng.integers(max(2 * C, 8), 3000))
        U = int(rng.integers(1, 4000))
        Ic = rsx.chunk_rows(I, C, c)
        base, rem = divmod(I, C)
        n_real = np.array([base + (k < rem) for k in range(C)])
        real_ids = np.concatenate([k * Ic + np.arange(n_real[k]) for k in range(C)])      # the relabelled ids that exist
        kind = trial % 4
        if kind == 0:
            degs = rng.integers(0, min(I, 6), U)
        elif kind == 1:
            degs = rng.integers(0, min(I, 120), U)
        elif kind == 2:
            degs = rng.integers(1, min(I, 30) + 1, U)
        else:
            degs = np.minimum(I - 1, (rng.pareto(1.0, U) * 3).astype(np.int64))
        rows = [np.sort(rng.choice(real_ids, int(g), replace=False)) for g in degs]
        owners = rng.random(U) < (0.05 if kind == 2 else 0.0)                          # users owning ALL items of range 0 (+ a few more)
        for uu in np.flatnonzero(owners):
            rows[uu] = np.unique(np.concatenate([np.arange(n_real[0]), rows[uu]]))
        indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        indices = (np.concatenate(rows) if indptr[-1] else np.zeros(0)).astype(np.int32)
        ip = torch.from_numpy(indptr).cuda()
        ix = torch.from_numpy(indices if len(indices) else np.zeros(1, np.int32)).cuda()
        B = U if trial % 3 == 0 else int(rng.integers(1, U + 1))
        epoch_pos = 0 if B == U else int(rng.integers(0, U - B + 1))
        cdf = rsx.build_item_cdf(ip, ix, C * Ic)
        sig = rsx.build_signature(ip, ix, c) if (trial % 2 and c) else None
        ws = torch.empty(rsx.bpr_sample_workspace(B, C * Ic), dtype=torch.uint8, device="cuda")
        outs = []
        for rep in range(2):
            u, i, j = (torch.full((B,), -7, dtype=torch.int32, device="cuda") for _ in range(3))
            cp = torch.full((C + 1,), -1, dtype=torch.int64, device="cuda")
            rsx.bpr_sample_chunked(ip, ix, C * Ic, I, C, B, 11, trial, epoch_pos, u, i, j, cp, c, 2 * trial + 1, ws, cdf, user_sig=sig)
            torch.cuda.synchronize()
            outs.append((u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy(), cp.cpu().numpy()))
        ctx = f"trial {trial}: U={U} I={I} C={C} c={c} B={B} kind={kind}"
        (ua, ia, ja, cpa), (ub, ib, jb, cpb) = outs
        live = ia >= 0
        n_live = int(live.sum())
        assert np.array_equal(cpa, cpb) and np.array_equal(ia, ib) and np.array_equal(ja, jb) and np.array_equal(ua[live], ub[live]), ctx
        assert len(np.unique(ua)) == B and np.all(ja[~live] == -1), ctx
        for p in np.flatnonzero(~live & (np.arange(B) >= cpa[-1])):                       # pairs without a positive come last --
            assert len(rows[ua[p]]) in (0, I), ctx                                        # an empty row, or one that owns the whole
            #                                                                               catalog (no negative exists: the plain sampler's
            #                                                                               rule too; found by the campaign at I = 21, 30)
        # cp counts the pairs that HAD a positive; a pair skipped for want of a negative keeps its place (i = -1 inside a range)
        assert cpa[0] == 0 and np.all(np.diff(cpa) >= 0) and cpa[-1] >= n_live, ctx
        inside = ia[:cpa[-1]]
        assert np.all(np.diff(inside[inside >= 0]) >= 0), ctx                             # ordered by item
        for k in range(C):
            sl = slice(cpa[k], cpa[k + 1])
            ok = ia[sl] >= 0
            assert np.all(ia[sl][ok] // Ic == k) and np.all(ja[sl][ok] // Ic == k), ctx
            assert np.all(ja[sl][ok] - k * Ic < n_real[k]), ctx                           # real rows, never padding
        for p in np.flatnonzero(live)[:: max(1, B // 300)]:
            row = rows[ua[p]]
            assert ia[p] in row and ja[p] not in row, ctx
        # a user who owns (nearly) every item of the range its positive fell in may end without a negative (192 draws, the last
        # 128 uniform over the range): skipped for this step, never served from another range
        for p in np.flatnonzero(~live & (np.arange(B) < cpa[-1])):
            row = rows[ua[p]]
            k = int(np.searchsorted(cpa, p, side="right") - 1)
            assert np.isin(k * Ic + np.arange(n_real[k]), row).mean() > 0.9, ctx


def test_exchange_range_callback_sees_every_range_in_order_and_its_errors_surface():
    """include/rsx.h: exchange_range -- one process, no process group: a chunked trainer with a callback that records what it is handed
    (range k, first row, rows, a stream handle) and does nothing else is the sharded schedule with an identity exchange: it must
    end at the tables of the unsharded chunked trainer on the same triplets; a callback that raises makes run() raise THAT exception
    (it must not unwind through the C frames, and nothing may hang)"""
    from recsys_pytorch_amd import rsx
    U, I, d, B, C = 20_000, 3_000, 64, 12_000, 3
    lr = resolvable_lr(B)

    def make():
        eng, P, Q, ip, ix = _engine(U, I, d, B, 8, C, lr, seed=9, hot=16)
        r = eng._build_relabel(ip, ix)
        eng._items_to_relabelled()
        return eng, P, r, ip

    def trainer(eng, P, r, ip, cb):
        return rsx.BPRTrainer(P, r["Q"], r["G"], ip, r["indices"], lr, B, seed=eng.seed, seed_key=eng.seed, neg_block=eng.neg_block,
                              hot=r["hot"], user_sig=r["sig"], item_cdf=r["cdf"], chunks=C, items_real=I, exchange_range=cb)

    eng, P, r, ip = make()
    plain = trainer(eng, P, r, ip, None)
    plain.run(3)
    torch.cuda.synchronize()
    plain.check()
    P_ref, Q_ref = P.clone(), r["Q"].clone()
    plain.close()

    calls = []
    eng2, P2, r2, ip2 = make()
    P0, Q0 = P2.clone(), r2["Q"].clone()
    tr = trainer(eng2, P2, r2, ip2, lambda k, first, rows, stream: calls.append((k, first, rows, stream != 0)))
    tr.run(3)
    torch.cuda.synchronize()
    tr.check()
    Ic = r2["Ic"]
    assert calls == [(k, k * Ic, Ic, True) for _ in range(3) for k in range(C)]          # every range of every step, in range order
    assert_update(P2.cpu().numpy(), P0.cpu().numpy(), P_ref.cpu().numpy(), "P (identity exchange_range vs no exchange)")
    assert_update(r2["Q"].cpu().numpy(), Q0.cpu().numpy(), Q_ref.cpu().numpy(), "Q (identity exchange_range vs no exchange)")
    assert float(r2["G"].abs().max()) == 0.0
    tr.close()

    class Boom(RuntimeError):
        pass

    def bad(k, first, rows, stream):
        if k == 1:
            raise Boom("range 1")
    eng3, P3, r3, ip3 = make()
    tr = trainer(eng3, P3, r3, ip3, bad)
    with pytest.raises(Boom):
        tr.run(1)
    torch.cuda.synchronize()                                 # what was queued before the failure drains: nothing hangs
    tr.close()
