"""Property tests (hypothesis) of the host arithmetic the sharded step rests on: who owns which user, how many rows an item range gets,
how items are dealt to ranges, which negative block is picked, how a sparse matrix is cut into segments.  CPU only."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

pytestmark = pytest.mark.filterwarnings("ignore")


@settings(max_examples=300, deadline=None)
@given(st.integers(0, 10_000_000), st.integers(1, 64))
def test_user_blocks_partition_the_users(U, W):
    """every user has exactly one owner, blocks are contiguous and in rank order, no block is larger than ceil(U / W)"""
    from recsys_pytorch_amd.sharded import user_block
    blocks = [user_block(U, r, W) for r in range(W)]
    assert blocks[0][0] == 0 and blocks[-1][1] == U
    per = -(-U // W) if U else 0
    for (a, b), (c, d) in zip(blocks, blocks[1:]):
        assert b == c
    assert all(0 <= b - a <= per for a, b in blocks) and sum(b - a for a, b in blocks) == U


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 5_000_000), st.integers(1, 8), st.integers(0, 16))
def test_chunk_rows_is_the_smallest_row_count_that_fits(I, C, nb):
    """include/rsx.h rsx_chunk_rows: every range holds ceil(I / C) real items, rounded up to whole negative blocks -- and not a block more"""
    from recsys_pytorch_amd import rsx
    Ic = rsx.chunk_rows(I, C, nb)
    need = -(-I // C)
    step = max(nb, 1)
    assert Ic >= need and Ic % step == 0 and Ic - need < step
    with pytest.raises(rsx.RsxError):
        rsx.chunk_rows(I, 9, nb)                        # RSX_MAX_CHUNKS = 8


@settings(max_examples=120, deadline=None)
@given(st.integers(2, 8), st.integers(0, 400), st.integers(0, 2**31 - 1), st.sampled_from(["zipf", "flat", "ties", "one giant"]))
def test_items_are_dealt_to_the_ranges_exactly_and_reproducibly(C, extra, seed, kind):
    """deal_items_to_ranges: every item gets a range, range k gets exactly cap[k] items, the same seed deals the same hands, another
    seed moves items (whenever there is anything to move), and no range carries more than its share of the heavy mass plus one item"""
    from recsys_pytorch_amd.sharded import deal_items_to_ranges
    I = C + extra
    rng = np.random.default_rng(seed)
    if kind == "zipf":
        mass = 1.0 / (1.0 + rng.permutation(I))
    elif kind == "flat":
        mass = np.ones(I)
    elif kind == "ties":
        mass = rng.integers(1, 4, I).astype(np.float64)
    else:
        mass = np.ones(I); mass[rng.integers(0, I)] = 10.0 * I
    base, rem = divmod(I, C)
    cap = np.array([base + (k < rem) for k in range(C)], dtype=np.int64)
    a = deal_items_to_ranges(mass, cap, np.random.default_rng(seed))
    b = deal_items_to_ranges(mass, cap, np.random.default_rng(seed))
    assert np.array_equal(a, b)
    assert a.min() >= 0 and a.max() < C and np.array_equal(np.bincount(a, minlength=C), cap)
    load = np.bincount(a, weights=mass, minlength=C)
    assert load.max() <= mass.sum() / C + mass.max() * (1 + 1e-9) + (mass.sum() / C) * 0.35     # balanced up to the heaviest item (+ slack of the random tail)
    if I >= 4 * C and kind != "flat":
        others = [deal_items_to_ranges(mass, cap, np.random.default_rng(seed + 1 + k)) for k in range(3)]
        assert any(not np.array_equal(a, o) for o in others)


@settings(max_examples=300, deadline=None)
@given(st.integers(2, 2_000_000), st.integers(1, 16), st.integers(64, 20_000), st.one_of(st.none(), st.integers(1, 20_000_000)), st.integers(2, 3))
def test_pick_neg_block_stays_in_its_bounds(I, max_block, slots, batch, min_block):
    from recsys_pytorch_amd.sharded import pick_neg_block
    c = pick_neg_block(I, max_block, slots, batch, min_block)
    assert 1 <= c <= max(1, max_block)
    if max_block >= 2:
        assert c >= 2
    if batch is not None and batch >= 10 * I and max_block >= 2 and min(min_block, max_block) * batch >= 20 * I:
        assert c == min(min_block, max_block)          # dense batches: the smallest admissible block


@settings(max_examples=150, deadline=None)
@given(st.lists(st.integers(0, 3000), min_size=1, max_size=300), st.integers(1, 1500))
def test_spmm_plan_covers_every_nonzero_exactly_once(degs, max_seg):
    """rsx_spmm_plan (host): the segments of a row tile it in order, none is longer than max_seg, empty rows get one empty segment
    or none -- whatever the row lengths"""
    import ctypes
    from recsys_pytorch_amd import rsx
    indptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    n = len(degs)
    L = rsx.lib()
    cnt = L.rsx_spmm_plan(indptr.ctypes.data, n, max_seg, None, None, None)
    assert cnt >= 0
    row, beg, ln = np.empty(cnt, np.int32), np.empty(cnt, np.int64), np.empty(cnt, np.int32)
    assert L.rsx_spmm_plan(indptr.ctypes.data, n, max_seg, row.ctypes.data, beg.ctypes.data, ln.ctypes.data) == cnt
    assert np.all(ln <= max_seg) and np.all(ln >= 0)
    covered = np.zeros(int(indptr[-1]), dtype=np.int32)
    for r, b, l in zip(row, beg, ln):
        assert indptr[r] <= b and b + l <= indptr[r + 1]
        covered[b:b + l] += 1
    assert np.all(covered == 1)
