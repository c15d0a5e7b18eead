"""world_size-2 gloo run of the user-sharded engine on CPU: two ranks on their own
triplets must equal one process on the concatenated batch (fp32 summation tolerance),
and the item replicas must stay identical (SURVEY section 8e parity check)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import UPDATE_TOL, assert_update, resolvable_lr  # noqa: E402
from rankpool import spawn  # noqa: E402  (persistent rank processes instead of a fresh interpreter per rank and test)  (lr = 0.05 * B: the update itself is compared)


def _worker(rank, world, port, P0, Q0, batches, lr, out, unique=False, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernels
    from recsys_pytorch_amd.sharded import BPREngine, user_block
    lo, hi = user_block(P0.shape[0], rank, world)
    P = torch.from_numpy(P0[lo:hi].copy())
    Q = torch.from_numpy(Q0.copy())
    eng = BPREngine(P, Q, lr, kernels=cpu_kernels, user_begin=lo, exchange=exchange)
    Q = eng.Q                                           # "scatter_gather" may re-home the item table (padding)
    losses, sums = [], []
    for (u, i, j) in batches:
        ul, il, jl = eng.route(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
        acc = eng.step(ul, il, jl, users_unique=unique)  # global batch size found by all-reduce
        losses.append(float(acc.sum()) / len(u))
        sums.append(eng.item_checksum())
    out[rank] = (lo, hi, P.numpy().copy(), Q.numpy().copy(), losses, sums)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("exchange", ["allreduce", "scatter_gather"])
def test_two_ranks_equal_one_process(oracle_mod, exchange):
    """exchange = all_reduce(G) + identical apply, or reduce_scatter(G) -> own item shard applied ->
    all_gather of the updated rows (97 items on 2 ranks: the padded-shard path)"""
    rng = np.random.default_rng(9)
    U, I, d, B, T = 301, 97, 64, 200, 5
    lr = resolvable_lr(B)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    batches = [(rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(T)]
    batches.append((rng.integers(0, 150, 64), rng.integers(0, I, 64), rng.integers(0, I, 64)))  # rank 1 idle
    single = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    ref_losses = [single.step(*b) for b in batches]
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 2000
    spawn(_worker, args=(world, port, P0, Q0, batches, lr, out, False, exchange), nprocs=world, join=True)
    P = np.zeros_like(P0)
    for r in range(world):
        lo, hi, Pr, Qr, losses, sums = out[r]
        P[lo:hi] = Pr
        assert np.allclose(losses, ref_losses, rtol=1e-5, atol=1e-6)
    assert np.array_equal(out[0][3], out[1][3]), "item replicas diverged"
    assert out[0][5] == out[1][5]
    assert_update(P, P0, single.P, "P")
    assert_update(out[0][3], Q0, single.Q, "Q")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("exchange", ["allreduce", "scatter_gather"])
def test_two_ranks_with_the_exchange_under_the_user_pass_equal_one_process(oracle_mod, exchange):
    """batches with unique users take the two-pass step (item pass -> async all-reduce of G ->
    user pass, BPREngine.overlap_exchange): same step as one launch"""
    rng = np.random.default_rng(10)
    U, I, d, B, T = 301, 97, 64, 200, 4
    lr = resolvable_lr(B)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    batches = [(rng.permutation(U)[:B], rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(T)]
    batches.append((rng.permutation(150)[:64], rng.integers(0, I, 64), rng.integers(0, I, 64)))   # rank 1 idle
    single = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    ref_losses = [single.step(*b) for b in batches]
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + 3) % 2000
    spawn(_worker, args=(2, port, P0, Q0, batches, lr, out, True, exchange), nprocs=2, join=True)
    P = np.zeros_like(P0)
    for r in range(2):
        lo, hi, Pr, Qr, losses, sums = out[r]
        P[lo:hi] = Pr
        assert np.allclose(losses, ref_losses, rtol=1e-5, atol=1e-6)
    assert np.array_equal(out[0][3], out[1][3]), "item replicas diverged"
    assert_update(P, P0, single.P, "P")
    assert_update(out[0][3], Q0, single.Q, "Q")


def _relabel_worker(rank, world, port, U, I, d, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernels
    from recsys_pytorch_amd.sharded import BPREngine
    rng = np.random.default_rng(100 + rank)                    # every rank its OWN users: different local popularity
    p = 1.0 / np.arange(1, I + 1) ** (0.8 + 0.3 * rank)
    p /= p.sum()
    rows = [np.sort(rng.choice(I, 12, replace=False, p=p)) for _ in range(U)]
    indptr = torch.arange(U + 1, dtype=torch.int64) * 12
    indices = torch.from_numpy(np.concatenate(rows).astype(np.int32))
    eng = BPREngine(torch.zeros(U, d), torch.randn(I, d, generator=torch.Generator().manual_seed(3)), 0.1, kernels=cpu_kernels,
                    user_begin=rank * U, seed=2020)
    eng.neg_block, eng.chunks = 6, 4
    eng.set_hot_items(torch.bincount(indices.long(), minlength=I), 16, 4)
    r = eng._build_relabel(indptr, indices)
    local_mass = np.bincount(indices.numpy(), minlength=I) / 12.0
    out[rank] = (r["rank_item"].numpy().copy(), r["item_rank"].numpy().copy(), int(r["Ic"]), local_mass, r["indices"].numpy().copy(),
                 indices.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_agree_on_the_relabelled_item_space():
    """the item-range pipelines train on a relabelled item space that must be THE SAME on every rank (the ranges' rows are
    all-reduced position by position) although every rank sees other users: the range assignment is a function of the sampling
    masses summed over the ranks.  Two gloo ranks with different local popularity: identical relabelling, a bijection onto the
    real rows, range masses balanced on the GLOBAL distribution, the local CSR relabelled consistently"""
    U, I, d, world = 3000, 1003, 8, 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + 41) % 2000
    spawn(_relabel_worker, args=(world, port, U, I, d, out), nprocs=world, join=True)
    (ri0, ir0, Ic, m0, im0, ix0), (ri1, ir1, _, m1, im1, ix1) = out[0], out[1]
    assert np.array_equal(ri0, ri1) and np.array_equal(ir0, ir1)
    assert np.array_equal(ri0[ir0], np.arange(I)) and (ri0 >= 0).sum() == I
    mass = m0 + m1
    share = np.array([mass[ri0[k * Ic:(k + 1) * Ic][ri0[k * Ic:(k + 1) * Ic] >= 0]].sum() for k in range(4)]) / mass.sum()
    assert np.abs(share - 0.25).max() < 0.01, share
    for im, ix in ((im0, ix0), (im1, ix1)):                     # each rank's CSR columns: the same items, relabelled, rows re-sorted
        assert np.array_equal(np.sort(ir0[ix].reshape(U, 12), axis=1), im.reshape(U, 12))


def _ranges_worker(rank, world, port, gpu, U, I, d, B, deg, chunks, steps, one_run, out, exchange="allreduce"):
    """one rank of the chunked native loop with the per-range exchange handed in as a callback (include/rsx.h:
    exchange_range) over gloo: the CPU stand-in trainer (gpu = False) or the HIP library, both ranks on the box's one GPU"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd.sharded import BPREngine
    dev = torch.device("cuda", 0) if gpu else torch.device("cpu")
    kernels = None
    if not gpu:
        import cpu_kernels as kernels
    rng = np.random.default_rng(500 + rank)                     # every rank its OWN users, with its own popularity skew
    p = 1.0 / np.arange(1, I + 1) ** (0.7 + 0.4 * rank)
    p /= p.sum()
    rows = [np.sort(rng.choice(I, deg, replace=False, p=p)) for _ in range(U)]
    indptr = (torch.arange(U + 1, dtype=torch.int64) * deg).to(dev)
    indices = torch.from_numpy(np.concatenate(rows).astype(np.int32)).to(dev)
    P = (torch.randn(U, d, generator=torch.Generator().manual_seed(100 + rank)) * 0.1).to(dev)
    Q = (torch.randn(I, d, generator=torch.Generator().manual_seed(7)) * 0.1).to(dev)
    P_init, Q_init = P.cpu().numpy().copy(), Q.cpu().numpy().copy()
    eng = BPREngine(P, Q, resolvable_lr(world * B), kernels=kernels, user_begin=rank * U, seed=11, exchange=exchange)
    assert eng.sharded and eng.comm is None and eng.exchange == exchange
    if gpu:
        assert (eng.set_neg_block(B, 8) > 0) == (B >= 2 * I)
        if B < 2 * I:
            eng.sorted_min_batch = 1          # the ranges without blocks (negatives over the real items of the positive's range)
        eng.set_hot_items(torch.bincount(indices.long(), minlength=I), 32, 4)
    else:
        eng.neg_block = 4
        eng.set_hot_items(torch.bincount(indices.long(), minlength=I), 8, 2)
    assert eng.set_chunks(chunks) == chunks
    acc = torch.zeros(eng.k.RSX_LOSS_SLOTS, dtype=torch.float32, device=dev)
    tr = eng.native_trainer(indptr, indices, B, loss_acc=acc)
    assert tr.chunks == chunks
    r = eng._relabel
    trips = []
    if one_run:                                                 # all steps queued by ONE call: the ranges of neighbouring steps interleave
        tr.run(steps, B, global_batch=world * B)
    else:
        for _ in range(steps):                                  # step by step, to dump what every step consumed
            tr.run(1, B, global_batch=world * B)
            if gpu:
                torch.cuda.synchronize()
            u, i, j = (t.cpu().numpy().astype(np.int64) for t in tr.last_batch()[:3])
            cp = tr.last_chunk_pos().cpu().numpy()
            live = i >= 0
            assert np.array_equal(np.searchsorted(i[live] // r["Ic"], np.arange(chunks + 1)), cp)   # range k = positions [cp[k], cp[k+1])
            assert np.all(i[live] // r["Ic"] == j[live] // r["Ic"])
            trips.append((u[live] + rank * U, i[live], j[live]))
    if gpu:
        torch.cuda.synchronize()
    lu, li, lj = (t.cpu().numpy().astype(np.int64) for t in tr.last_batch()[:3])       # what the LAST step consumed (either arm)
    last = (lu[li >= 0] + rank * U, li[li >= 0], lj[li >= 0])
    Qm = r["Q"].cpu().numpy().copy()                            # the relabelled replica, padding rows included
    eng.adopt(tr)                                               # checks the run (no triplet left its range), item rows back to the caller's ids
    tr.close()
    if exchange == "direct":
        assert eng._mesh[0].info() == (rank, world, steps * chunks)       # one exchange per item range and step went through the mesh
    eng.close_mesh()                                            # (exchange = "direct": collective -- checks that no wait gave up, barrier, unmap)
    out[rank] = (P.cpu().numpy(), eng.Q.cpu().numpy(), Qm, trips, r["rank_item"].cpu().numpy(), float(acc.sum()), P_init, Q_init,
                 float(r["G"].abs().max()), last)
    dist.barrier()
    dist.destroy_process_group()


def _check_ranges(oracle_mod, world, port, gpu, U, I, d, B, deg, chunks, steps, exchange="allreduce"):
    """two ranks on the chunked native loop == ONE process (the oracle) on the concatenation of what the ranks sampled"""
    mgr = mp.Manager()
    res = {}
    for one_run in (False, True):
        out = mgr.dict()
        spawn(_ranges_worker, args=(world, port + 3 * one_run, gpu, U, I, d, B, deg, chunks, steps, one_run, out, exchange), nprocs=world, join=True)
        res[one_run] = [out[r] for r in range(world)]
    for arm in res.values():
        for r in range(1, world):
            assert np.array_equal(arm[0][2], arm[r][2]), f"relabelled item replicas diverged (some range), rank {r}"
            assert np.array_equal(arm[0][1], arm[r][1]), f"item replicas diverged, rank {r}"
            assert np.array_equal(arm[0][4], arm[r][4])
        assert all(a[8] == 0.0 for a in arm)                      # every range's gradient rows were applied and cleared
    step_arm = res[False]
    rank_item, Q_init = step_arm[0][4], step_arm[0][7]
    P_init = np.concatenate([step_arm[r][6] for r in range(world)])
    lr = resolvable_lr(world * B)
    orc = oracle_mod.MFOracle(P_init, Q_init, "sgd", lr)
    loss_sum = 0.0
    for t in range(steps):
        u, i, j = (np.concatenate([step_arm[r][3][t][c] for r in range(world)]) for c in range(3))
        assert len(u) == world * B                              # (every user of these CSRs has a usable row: 1 / sum_r B_r is exact)
        loss_sum += orc.step(u, rank_item[i], rank_item[j]) * len(u)          # the triplets in the caller's item ids
    assert abs(sum(a[5] for a in step_arm) - loss_sum) < 1e-5 * loss_sum
    P = np.concatenate([a[0] for a in step_arm])
    assert_update(P, P_init, orc.P, "P (step by step)")
    assert_update(step_arm[0][1], Q_init, orc.Q, "Q (step by step)")
    # the same steps queued by one call (same seeds, same triplets): the same tables
    for rr in range(world):
        for x, y, what in zip(res[True][rr][9], step_arm[rr][3][-1], "uij"):
            assert np.array_equal(x, y), f"rank {rr}: the last step of the single call consumed other triplets ({what}) than the same step run alone"
    assert_update(np.concatenate([a[0] for a in res[True]]), P_init, P, "P (one run vs step by step)")
    assert_update(res[True][0][1], Q_init, step_arm[0][1], "Q (one run vs step by step)")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("I,chunks", [(203, 2), (150, 3)])
def test_two_ranks_item_ranges_with_range_callbacks_equal_one_process(oracle_mod, I, chunks):
    """the N > 1 DEFAULT schedule of bench.py -- every step as pipelines over item ranges, the exchange range by range -- with
    TWO ranks: the engine's host logic over gloo on the CPU stand-in trainer (tests/cpu_kernels.py)"""
    _check_ranges(oracle_mod, 2, 29500 + (os.getpid() + 53 + I) % 2000, False, 500, I, 32, 450, 6, chunks, 3)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("I,chunks,d", [(2500, 2, 64), (1999, 3, 128), (7001, 2, 128)])
def test_two_ranks_item_ranges_on_hip_kernels_equal_one_process(oracle_mod, I, chunks, d):
    """the same through csrc/rsx_train.hip's chunked loop and the HIP kernels, two processes sharing the box's GPU: per-range
    collectives in the same order on ranks whose range kernels finish at different times, sub-buffers G + lo * d with padding rows
    summing two ranks' partials, 1 / sum_r B_r, replica identity per range -- step by step against the oracle on the concatenated
    triplets, and the same steps queued by ONE rsx_bpr_trainer_run.  I = 7001: B < 2 I, the ranges without blocks (the form of
    BASELINE configs[3])"""
    _check_ranges(oracle_mod, 2, 29500 + (os.getpid() + 59 + I) % 2000, True, 9000, I, d, 6000, 10, chunks, 4)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("I,chunks,d", [(2500, 2, 64), (1999, 3, 128), (7001, 2, 128)])
def test_two_ranks_item_ranges_over_the_direct_mesh_equal_one_process(oracle_mod, I, chunks, d):
    """the same schedule with the library's OWN exchange (include/rsx.h: rsx_mesh_*, exchange = "direct"): two processes on the
    box's GPU map each other's item table, gradient buffer and mailbox over HIP IPC; per item range every rank sums ITS slice by
    reading the peer's rows directly, applies it, and copies the other slice from its owner -- no torch.distributed collective in
    the step (gloo only carries the handles and the barriers around the run).  Step by step against the oracle on the concatenated
    triplets and all steps queued by one call, replicas identical row by row, G zero, no wait gave up, steps x ranges exchanges"""
    _check_ranges(oracle_mod, 2, 29500 + (os.getpid() + 83 + I) % 2000, True, 9000, I, d, 6000, 10, chunks, 4, exchange="direct")


@pytest.mark.timeout(600)
def test_ranks_item_ranges_on_random_shapes(oracle_mod):
    """three random (ranks, users, items, ranges, batch) problems through the same check (the CPU stand-in trainer over gloo)"""
    from conftest import fuzz
    rng, trials = fuzz(77, 3)
    for trial in range(trials):
        world = int(rng.integers(2, 4))
        chunks = int(rng.integers(2, 5))
        I = int(rng.integers(8 * chunks, 260))
        B = int(rng.integers(2 * I, 3 * I + 40))                     # (the blocked layout: B >= 2 I on every rank)
        U = B + int(rng.integers(0, 200))
        _check_ranges(oracle_mod, world, 29500 + (os.getpid() + 151 + 17 * trial) % 2000, False, U, I, int(rng.choice([8, 32])), B, 5, chunks, 2)


@pytest.mark.timeout(600)
def test_three_ranks_item_ranges_with_range_callbacks_equal_one_process(oracle_mod):
    """... and with THREE ranks (1 / sum_r B_r over three batches, three user blocks, every replica against rank 0's)"""
    _check_ranges(oracle_mod, 3, 29500 + (os.getpid() + 61) % 2000, False, 400, 171, 32, 360, 6, 2, 3)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,I,chunks,d", [(3, 1999, 3, 128), (4, 2500, 2, 64)])
def test_more_ranks_item_ranges_over_the_direct_mesh_equal_one_process(oracle_mod, world, I, chunks, d):
    """the direct mesh with three and four ranks (processes on the box's GPU): every rank reads two / three peers' rows for its slice and
    gathers from two / three owners; slices with a remainder.  Same checks as with two ranks"""
    _check_ranges(oracle_mod, world, 29500 + (os.getpid() + 89 + I) % 2000, True, 6000, I, d, 4000, 10, chunks, 3, exchange="direct")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_ranks_item_ranges_on_hip_kernels_on_random_shapes(oracle_mod):
    """three random (ranks, exchange, items, ranges, row width, batch above or below two triplets per item) problems through the same
    check on the HIP kernels (processes on the box's GPU)"""
    from conftest import fuzz
    rng, trials = fuzz(88, 3)
    for trial in range(trials):
        world = int(rng.integers(2, 4))
        exchange = ["direct", "allreduce", "direct"][trial % 3]
        chunks = int(rng.integers(2, 5))
        d = int(rng.choice([32, 64, 128, 256]))
        I = int(rng.integers(200, 6000))
        B = int(rng.integers(I // 2 + 200, 3 * I + 500))
        U = B + int(rng.integers(0, 2000))
        _check_ranges(oracle_mod, world, 29500 + (os.getpid() + 163 + 19 * trial) % 2000, True, U, I, d, B, int(rng.integers(4, 12)), chunks, 3,
                      exchange=exchange)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("I,chunks,exchange", [(2500, 2, "allreduce"), (9001, 3, "allreduce"), (2500, 2, "direct"), (9001, 3, "direct")])
def test_two_ranks_item_ranges_soak(I, chunks, exchange):
    """sixty steps of the range schedule queued by ONE call on each of two ranks (HIP loop, two processes on the box's GPU, the
    exchange range by range over gloo): the pipelines of neighbouring steps interleave for a long stretch -- afterwards the item
    replicas are identical row by row (padding rows included), every gradient row is applied and cleared, no triplet left its
    range (adopt() checks), everything finite, and the model has learned (the summed loss of the last steps is below ln 2).
    I = 9001: below two triplets per item, the ranges without blocks"""
    world, U, d, B, steps = 2, 9000, 64, 6000, 60
    mgr = mp.Manager()
    out = mgr.dict()
    spawn(_ranges_worker, args=(world, 29500 + (os.getpid() + 71 + I + 7 * len(exchange)) % 2000, True, U, I, d, B, 10, chunks, steps, True, out, exchange),
             nprocs=world, join=True)
    a, b = out[0], out[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1]), "item replicas diverged"
    assert a[8] == 0.0 and b[8] == 0.0
    for r in (a, b):
        assert np.isfinite(r[0]).all() and np.isfinite(r[1]).all()
        assert np.abs(r[0] - r[6]).max() > 1e-3 and np.abs(r[1] - r[7]).max() > 1e-3           # the tables moved
    mean_loss = (a[5] + b[5]) / (world * B * steps)
    assert 0.0 < mean_loss < 0.6931, mean_loss


def _gpu_worker(rank, world, port, P0, Q0, batches, lr, out, unique=False, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd.sharded import BPREngine, user_block
    dev = torch.device("cuda", 0)                      # both ranks share the one GPU of the test box
    lo, hi = user_block(P0.shape[0], rank, world)
    P = torch.from_numpy(P0[lo:hi].copy()).to(dev)
    Q = torch.from_numpy(Q0.copy()).to(dev)
    eng = BPREngine(P, Q, lr, user_begin=lo, exchange=exchange)     # default kernels: the HIP library
    Q = eng.Q
    if unique:
        eng.set_hot_items(torch.bincount(torch.from_numpy(np.concatenate([b[1] for b in batches])), minlength=Q0.shape[0]), 16, 4)
    losses = []
    for s, (u, i, j) in enumerate(batches):
        ul, il, jl = eng.route(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(j).to(dev))
        if unique:     # two-pass step; odd steps through the blocked kernel, even ones through the plain one
            acc = eng.step(ul, il, jl, users_unique=True, neg_block=8 * (s % 2), neg_key=77 * (s % 2))
        else:
            acc = eng.step(ul, il, jl)
        losses.append(float(acc.sum()) / len(u))
    torch.cuda.synchronize()
    if exchange == "direct":
        assert eng._mesh[0].info() == (rank, world, len(batches))
    eng.close_mesh()
    out[rank] = (lo, hi, P.cpu().numpy(), Q.cpu().numpy(), losses)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["allreduce", "scatter_gather", "direct"])
def test_two_ranks_on_hip_kernels_equal_one_process(oracle_mod, exchange):
    """same check with the real HIP kernels: two processes (sharing the box's GPU, gloo for the
    all-reduce) on their own triplets == one process on the concatenated batch"""
    rng = np.random.default_rng(19)
    U, I, d, B, T = 4001, 1500, 128, 3000, 4
    lr = resolvable_lr(B)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    batches = [(rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(T)]
    single = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    ref_losses = [single.step(*b) for b in batches]
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + 7) % 2000
    spawn(_gpu_worker, args=(2, port, P0, Q0, batches, lr, out, False, exchange), nprocs=2, join=True)
    P = np.zeros_like(P0)
    for r in range(2):
        lo, hi, Pr, Qr, losses = out[r]
        P[lo:hi] = Pr
        assert np.allclose(losses, ref_losses, rtol=1e-5, atol=1e-6)
    assert np.array_equal(out[0][3], out[1][3]), "item replicas diverged"
    assert_update(P, P0, single.P, "P")
    assert_update(out[0][3], Q0, single.Q, "Q")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_ranks_routed_steps_on_random_shapes(oracle_mod):
    """four random (ranks 2-4, exchange, users, items -- fewer than ranks too --, row width, batch, unique users or repeats) problems: every
    rank steps on the triplets routed to its user block (BPREngine.route / step), the ranks together equal ONE process on the whole
    batch, the item replicas are identical"""
    from conftest import fuzz
    rng, trials = fuzz(4040, 4)
    for trial in range(trials):
        world = int(rng.integers(2, 5))
        exchange = ["allreduce", "scatter_gather", "direct"][trial % 3]
        unique = bool(trial % 2)
        d = int(rng.choice([32, 64, 128, 256]))
        U, I = int(rng.integers(world, 3000)), int(rng.integers(2, 2500))
        B = int(rng.integers(1, U + 1)) if unique else int(rng.integers(1, 3000))
        T = 3
        lr = resolvable_lr(B)
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        batches = [((rng.permutation(U)[:B] if unique else rng.integers(0, U, B)), rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(T)]
        ctx = f"trial {trial}: world={world} {exchange} unique={unique} U={U} I={I} d={d} B={B}"
        single = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
        ref_losses = [single.step(*b) for b in batches]
        mgr = mp.Manager()
        out = mgr.dict()
        spawn(_gpu_worker, args=(world, 29500 + (os.getpid() + 173 + 23 * trial) % 2000, P0, Q0, batches, lr, out, unique, exchange), nprocs=world, join=True)
        P = np.zeros_like(P0)
        for r in range(world):
            lo, hi, Pr, Qr, losses = out[r]
            P[lo:hi] = Pr
            assert np.allclose(losses, ref_losses, rtol=2e-5, atol=1e-6), ctx
            assert np.array_equal(out[0][3], Qr), ctx + f": item replica of rank {r} diverged"
        assert_update(P, P0, single.P, "P, " + ctx)
        assert_update(out[0][3], Q0, single.Q, "Q, " + ctx)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["allreduce", "scatter_gather", "direct"])      # ("direct": one pass, the mesh exposed)
def test_two_ranks_on_hip_kernels_with_the_exchange_under_the_user_pass(oracle_mod, exchange):
    rng = np.random.default_rng(29)
    U, I, d, B, T = 4001, 1501, 128, 3000, 4
    lr = resolvable_lr(B)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    batches = [(rng.permutation(U)[:B], (rng.integers(0, I, B) ** 2) // I, rng.integers(0, I, B)) for _ in range(T)]
    single = oracle_mod.MFOracle(P0, Q0, "sgd", lr)
    ref_losses = [single.step(*b) for b in batches]
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + 11) % 2000
    spawn(_gpu_worker, args=(2, port, P0, Q0, batches, lr, out, True, exchange), nprocs=2, join=True)
    P = np.zeros_like(P0)
    for r in range(2):
        lo, hi, Pr, Qr, losses = out[r]
        P[lo:hi] = Pr
        assert np.allclose(losses, ref_losses, rtol=1e-5, atol=1e-6)
    assert np.array_equal(out[0][3], out[1][3]), "item replicas diverged"
    assert_update(P, P0, single.P, "P")
    assert_update(out[0][3], Q0, single.Q, "Q")


def _gpu_sampled_worker(rank, world, port, mode, U, I, d, B, steps, out, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    dev = torch.device("cuda", 0)                      # both ranks share the one GPU of the test box
    ip, ix = synthetic_csr(U, I, 10, dev, seed=40 + rank)          # this rank's block of users
    torch.manual_seed(100 + rank)
    P = torch.randn(U, d, device=dev) * 0.1
    torch.manual_seed(7)
    Q = torch.randn(I, d, device=dev) * 0.1
    P_init, Q_init = P.cpu().numpy(), Q.cpu().numpy()
    eng = BPREngine(P, Q, resolvable_lr(world * B), user_begin=rank * U, seed=11, exchange=exchange)
    Q = eng.Q
    eng.set_neg_block(B, 8)
    eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 32, 4)
    if mode == "python":
        for _ in range(steps):
            eng.sampled_step_overlapped(ip, ix, B, global_batch=world * B, want_loss=False)
    else:
        tr = eng.native_trainer(ip, ix, B)
        tr.run(steps, B, global_batch=world * B)
        torch.cuda.synchronize()
        eng.adopt(tr)
        tr.close()
    torch.cuda.synchronize()
    eng.close_mesh()
    # (the Python-driven engine's epoch_pos already counts the batch it sampled ahead)
    pos = eng.epoch_pos if mode == "native" else eng._bufs[eng._cur]["pos_before"]
    out[(mode, rank)] = (P.cpu().numpy(), Q.cpu().numpy(), eng.step_count, pos, P_init, Q_init)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("B,I,exchange", [(6000, 2500, "allreduce"), (3000, 4000, "allreduce"), (6000, 2501, "scatter_gather"),
                                          (6000, 2503, "direct")])
def test_two_ranks_native_loop_with_exchange_callbacks_equals_python_driven(B, I, exchange):
    """user-sharded SAMPLED steps, two processes on the HIP kernels: the native loop (exchange handed
    in as callbacks, all-reduce of G under the user pass) against the Python-driven engine on the
    same triplets; item replicas identical across the ranks in both"""
    U, d, steps = 9000, 64, 5
    mgr = mp.Manager()
    out = mgr.dict()
    for k, mode in enumerate(("python", "native")):
        port = 29500 + (os.getpid() + 13 + 17 * k + B) % 2000
        spawn(_gpu_sampled_worker, args=(2, port, mode, U, I, d, B, steps, out, exchange), nprocs=2, join=True)
    for mode in ("python", "native"):
        assert np.array_equal(out[(mode, 0)][1], out[(mode, 1)][1]), f"item replicas diverged ({mode})"
    for r in range(2):
        Pp, Qp, sp_, pp, P_init, Q_init = out[("python", r)]
        Pn, Qn, sn, pn = out[("native", r)][:4]
        assert (sp_, pp) == (sn, pn) and sn == steps
        assert_update(Pn, P_init, Pp, "P")          # the five-step updates agree to 1e-5 of their size
        assert_update(Qn, Q_init, Qp, "Q")


def _rccl_one_rank_worker(rank, port, U, I, d, B, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # "nccl" IS RCCL on ROCm
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    ip, ix = synthetic_csr(U, I, 10, dev, seed=40)
    lib_comm = rsx.Comm()                  # RCCL communicator owned by librsx (include/rsx.h: rsx_comm_*), one rank

    init = {}

    def run(force, exchange, two_pass, mode, comm=None, chunks=0):
        torch.manual_seed(100)
        P = torch.randn(U, d, device=dev) * 0.1
        torch.manual_seed(7)
        Q = torch.randn(I, d, device=dev) * 0.1
        init.setdefault("tables", (P.cpu().numpy(), Q.cpu().numpy()))
        eng = BPREngine(P, Q, resolvable_lr(B), seed=11, exchange=exchange, force_sharded=force, comm=comm)
        eng.overlap_exchange = bool(two_pass) and force
        eng.set_neg_block(B, 8)
        eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 32, 4)
        if chunks:
            eng.set_chunks(chunks)
        if mode == "python":
            for _ in range(steps):
                eng.sampled_step_overlapped(ip, ix, B, global_batch=B, want_loss=False)
        else:
            tr = eng.native_trainer(ip, ix, B)
            assert (tr.chunks > 1) == bool(chunks)
            tr.run(2, B, global_batch=B)
            tr.run(steps - 2, B, global_batch=B)
            torch.cuda.synchronize()
            eng.adopt(tr)                       # (chunked: checks the run and copies the item rows back)
            tr.close()
        torch.cuda.synchronize()
        return P.cpu().numpy(), eng.Q.cpu().numpy()

    def run_stale(exchange, lr=resolvable_lr(B) / 10, comm=None):
        """native loop with the opt-in one-step-stale exchange, and the same recurrence driven by hand on the same
        triplets: step t's kernel reads the item table WITHOUT the update of step t-1 (applied right after it)"""
        from recsys_pytorch_amd import rsx
        def tables():
            torch.manual_seed(100)
            P = torch.randn(U, d, device=dev) * 0.1
            torch.manual_seed(7)
            return P, torch.randn(I, d, device=dev) * 0.1
        P, Q = tables()
        eng = BPREngine(P, Q, lr, seed=11, exchange=exchange, force_sharded=True, comm=comm)
        eng.stale_exchange = True
        eng.set_neg_block(B, 8)
        eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 32, 4)
        tr = eng.native_trainer(ip, ix, B)
        tr.run(2, B, global_batch=B)            # two calls: every run drains its last exchange
        tr.run(steps - 2, B, global_batch=B)
        torch.cuda.synchronize()
        tr.close()
        assert not eng._pending

        def by_hand(stale):
            P2, Q2 = tables()
            ref = BPREngine(P2, Q2, lr, seed=11)
            ref.set_neg_block(B, 8)
            ref.set_hot_items(torch.bincount(ix.long(), minlength=I), 32, 4)
            G = [ref.G, torch.zeros_like(ref.G)]
            for lo, hi in ((0, 2), (2, steps)):
                for t in range(hi - lo):
                    u, i, j = ref.sample(ip, ix, B)
                    rsx.bpr_step(P2, Q2, G[t & 1], u, i, j, lr, 1.0 / B, users_unique=True, hot=ref.hot,
                                 neg_block=ref.neg_block, neg_key=ref.last_neg_key)
                    rsx.fold_hot_grad(G[t & 1], ref.hot)
                    ref.step_count += 1
                    if not stale:
                        rsx.apply_item_grad(Q2, G[t & 1], lr)           # the synchronous step
                    elif t > 0:
                        rsx.apply_item_grad(Q2, G[(t - 1) & 1], lr)     # after the NEXT step's kernel
                if stale:
                    rsx.apply_item_grad(Q2, G[(hi - lo - 1) & 1], lr)   # end of a run: drained
            torch.cuda.synchronize()
            return P2.cpu().numpy(), Q2.cpu().numpy()

        return (P.cpu().numpy(), eng.Q.cpu().numpy()), by_hand(True), by_hand(False)

    res = {"plain": run(False, "allreduce", False, "native")}
    for exchange in ("allreduce", "scatter_gather"):
        res[("stale", exchange)], res[("stale_ref", exchange)], res[("stale_sync", exchange)] = run_stale(exchange)
    for exchange in ("allreduce", "scatter_gather"):
        for two_pass in (False, True):
            for mode in ("native", "python"):
                res[(exchange, two_pass, mode)] = run(True, exchange, two_pass, mode)
            # the same schedules with the exchange issued BY THE LIBRARY (RCCL from librsx on the trainer's own stream)
            res[(exchange, two_pass, "native, library RCCL")] = run(True, exchange, two_pass, "native", comm=lib_comm)
    res[("stale", "library RCCL")], _, _ = run_stale("allreduce", comm=lib_comm)
    # the step as a pipeline over item ranges: one GPU (apply range by range) and sharded with the library's RCCL (all-reduce
    # + apply range by range) draw the same triplets and must end at the same tables
    ck = 4 if B >= 2 * I else 0            # (the ranges need blocked negatives: two triplets per item and step)
    res[("chunked", "one GPU")] = run(False, "allreduce", False, "native", chunks=ck)
    res[("chunked", "library RCCL")] = run(True, "allreduce", False, "native", comm=lib_comm, chunks=ck)
    lib_comm.close()
    res["init"] = init["tables"]
    out.update(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("B,I", [(6000, 2500), (3000, 4001)])
def test_exchange_over_rccl_with_one_rank_equals_the_unsharded_step(B, I):
    """the exchange path on the REAL collective library: a process group of one rank over RCCL (all a
    one-GPU box offers), BPREngine(force_sharded=True).  all_reduce / reduce_scatter + all_gather over one
    rank are identities, so every schedule (exchange kind x one/two-pass x native loop / Python-driven)
    must reproduce the unsharded step: what this checks is the ordering between the trainer's
    streams, the collective's stream and the callbacks (a collective that started before G was complete,
    or an apply that ran before the collective ended, shows up as a different table)"""
    U, d, steps = 9000, 64, 5
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + 29 + B) % 2000
    spawn(_rccl_one_rank_worker, args=(port, U, I, d, B, steps, out), nprocs=1, join=True)
    P0, Q0 = out["plain"]
    P_init, Q_init = out["init"]
    assert len(out) == 23
    from conftest import delta_err
    bad = {}
    for key, (P, Q) in out.items():
        if key == "init" or key[0] in ("stale", "stale_ref", "stale_sync", "chunked"):
            continue
        eP, eQ = delta_err(P, P_init, P0), delta_err(Q, Q_init, Q0)    # the five-step updates agree to 1e-5 of their size
        if eP > UPDATE_TOL or eQ > UPDATE_TOL:
            bad[key] = (eP, eQ)
    assert not bad, f"update error (P, Q) per schedule: {bad}"
    (Pc, Qc), (Pr, Qr) = out[("chunked", "one GPU")], out[("chunked", "library RCCL")]
    assert_update(Pr, P_init, Pc, "P chunked, library RCCL vs one GPU")
    assert_update(Qr, Q_init, Qc, "Q chunked, library RCCL vs one GPU")
    if B >= 2 * I:
        assert np.abs(Qc - Q0).max() > 1e-3 * np.abs(Q0 - Q_init).max()  # (other triplets than the unchunked layout)
    # the opt-in one-step-stale exchange: equal to its own recurrence driven by hand, and NOT the synchronous step
    for exchange in ("allreduce", "scatter_gather"):
        (P, Q), (Pr, Qr), (Ps, Qs) = out[("stale", exchange)], out[("stale_ref", exchange)], out[("stale_sync", exchange)]
        if exchange == "allreduce":          # the library's own RCCL exchange follows the same recurrence
            Pl, Ql = out[("stale", "library RCCL")]
            assert np.abs(Pl - Pr).max() < 1e-5 * np.abs(Qr - Q_init).max() and np.abs(Ql - Qr).max() < 1e-5 * np.abs(Qr - Q_init).max()
        scale = np.abs(Qr - Q_init).max()      # the five updates are O(0.1) of the table
        assert scale > 1e-2
        assert np.abs(P - Pr).max() < 1e-5 * scale and np.abs(Q - Qr).max() < 1e-5 * scale, exchange
        assert np.abs(Q - Qs).max() > 1e-3 * scale, "the stale recurrence cannot be told from the synchronous step"


def _mesh_timeout_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd import rsx
    dev = torch.device("cuda", 0)
    Q, G = rsx.mesh_tensor(1000, 64), rsx.mesh_tensor(1000, 64)       # (memory the library has already exported: include/rsx.h rsx_mesh_alloc)
    Q.fill_(1.0); G.fill_(float(rank + 1))
    mesh = rsx.Mesh(Q, G)
    mesh.set_wait_limit(0.5)
    # a healthy exchange first: Q -= 0.5 * (1 + 2) on both ranks, G cleared
    mesh.exchange_apply(0, 1000, 0.5)
    mesh.check()
    ok = bool((Q == -0.5).all()) and float(G.abs().max()) == 0.0
    dist.barrier()
    err = None
    if rank == 0:                # ... then rank 1 never shows up for the next one: rank 0's waits give up after 0.5 s, and say so
        G.fill_(1.0)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); mesh.exchange_apply(0, 1000, 0.5); t1.record()
        try:
            mesh.check()
        except rsx.RsxError as e:
            err = str(e)
        out["ms"] = t0.elapsed_time(t1)
    out[rank] = (ok, err)
    mesh.close()                 # (collective: barrier inside)
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_mesh_exchange_and_a_missing_peer_is_an_error_not_a_hang():
    """rsx_mesh on its own (two processes on the box's GPU): one exchange sums both ranks' gradients into both replicas; when a peer never
    queues its half of the next exchange, the waiting rank's kernels give up at the mesh's limit and rsx_mesh_check reports it -- wrong
    rows, loudly, never a hung GPU"""
    mgr = mp.Manager()
    out = mgr.dict()
    spawn(_mesh_timeout_worker, args=(2, 29500 + (os.getpid() + 97) % 2000, out), nprocs=2, join=True)
    assert out[0][0] and out[1][0]                                   # the healthy exchange: both replicas updated, G zero
    assert out[1][1] is None and out[0][1] is not None and "gave up waiting" in out[0][1]
    assert 400.0 < out["ms"] < 5000.0                                # two waits of 0.5 s each, not a hang


def _mesh_shapes_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd import rsx
    dev = torch.device("cuda", 0)
    res = []
    for rows, d in ((1, 32), (5, 64), (257, 128), (1000, 256)):          # fewer rows than ranks, odd slices, every row width
        gen = torch.Generator().manual_seed(1000 * rows + d)
        Q0 = torch.randn(rows, d, generator=gen)
        Gs = [torch.randn(rows, d, generator=gen) for _ in range(world)]      # every rank's partial sums (the same draws on all ranks)
        # the exchanged tables live in memory the library has already exported (rsx_mesh_alloc): twice the runtime refused to export a
        # pooled allocation of torch's allocator in a process that had mapped peers' memory before (GPUTEST_r05, profiles/r06_mesh_stress.txt);
        # the borrowed-memory path has a test of its own below
        Q, G = rsx.mesh_tensor(rows, d), rsx.mesh_tensor(rows, d)
        Q.copy_(Q0); G.copy_(Gs[rank])
        mesh = rsx.Mesh(Q, G)
        assert mesh.export_retries() == 0
        mesh.set_wait_limit(20.0)                                               # (a lost peer fails the test, it does not hang the box)
        # two exchanges over parts of the table, then the whole: rows [0, rows / 2), [rows / 2, rows) -- like item ranges -- and a second
        # step over everything with fresh partial sums
        half = rows // 2
        if half > 0:
            mesh.exchange_apply(0, half, 0.5)
        mesh.exchange_apply(half, rows - half, 0.5)
        mesh.check()
        want = Q0 - 0.5 * sum(Gs)
        ok1 = bool(torch.allclose(Q.cpu(), want, rtol=1e-5, atol=1e-5)) and float(G.abs().max()) == 0.0
        G.copy_(Gs[(rank + 1) % world].to(dev))                                 # step 2: the partial sums handed round
        mesh.exchange_apply(0, rows, 0.25)
        mesh.check()
        want2 = want - 0.25 * sum(Gs)
        ok2 = bool(torch.allclose(Q.cpu(), want2, rtol=1e-5, atol=1e-5)) and float(G.abs().max()) == 0.0
        res.append((rows, d, ok1, ok2, Q.cpu().numpy().copy(), mesh.info()[2]))
        mesh.close()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def _mesh_borrowed_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd import rsx
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(5)
    Q0 = torch.randn(300, 64, generator=gen)
    Gs = [torch.randn(300, 64, generator=gen) for _ in range(world)]
    Q, G = Q0.clone().to(dev), Gs[rank].clone().to(dev)            # torch's own (pooled) memory
    res = None
    try:
        mesh = rsx.Mesh(Q, G)
    except rsx.RsxError as e:                                        # (collective outcome: every rank raises the same list)
        res = ("refused", str(e))
    else:
        mesh.exchange_apply(0, 300, 0.5)
        mesh.check()
        res = ("ok", bool(torch.allclose(Q.cpu(), Q0 - 0.5 * sum(Gs), rtol=1e-5, atol=1e-5)) and float(G.abs().max()) == 0.0)
        mesh.close()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_mesh_over_borrowed_memory_works_or_says_exactly_why():
    """rsx_mesh_local over tables in the CALLER's memory (torch's pooled allocations): validated, exported with bounded retries.  On this
    runtime the export of such an allocation has been refused twice, persistently, in processes that had mapped peers' memory before --
    then every rank must get the same error, and it must name the call, the allocation and the way out (rsx_mesh_alloc); otherwise the
    exchange must be right"""
    mgr = mp.Manager()
    out = mgr.dict()
    spawn(_mesh_borrowed_worker, args=(2, 29500 + (os.getpid() + 211) % 2000, out), nprocs=2, join=True)
    assert out[0][0] == out[1][0]
    if out[0][0] == "refused":
        for r in (0, 1):
            assert "hipIpcGetMemHandle(allocation of" in out[r][1] and "rsx_mesh_alloc" in out[r][1], out[r][1]
    else:
        assert out[0][1] and out[1][1]


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_mesh_exchange_on_small_and_odd_tables(world):
    """rsx_mesh_exchange_apply on its own, with two, three and four ranks (processes on the box's GPU -- more than one peer to read, slices
    with a remainder): tables with fewer rows than ranks, odd slice boundaries, every row width (32 .. 256), part of a table and the whole
    of it: Q -= lr * (sum of the ranks' G) on every rank, G zero afterwards, the replicas bit-identical"""
    mgr = mp.Manager()
    out = mgr.dict()
    spawn(_mesh_shapes_worker, args=(world, 29500 + (os.getpid() + 101 + 13 * world) % 2000, out), nprocs=world, join=True)
    for r in range(1, world):
        for a, b in zip(out[0], out[r]):
            assert a[:4] == b[:4] and a[2] and a[3], (r, a[:4], b[:4])
            assert np.array_equal(a[4], b[4]), ("replicas differ", r, a[0], a[1])
            assert a[5] == b[5] == (3 if a[0] > 1 else 2)
