"""Interaction data for the BPR-MF path: CSR holders and the synthetic workload.

The reference's dataset/split code (data/dataset.py, data/preprocess.py) is out
of scope; what the hot path needs from it is only
`.num_users`, `.num_items`, `.train_data` (scipy CSR) and the eval input/target
matrices (models/MF.py:16-17,45; main.py:62-63).  `InteractionData` is that
duck type; `synthetic_csr` builds the BASELINE.json workloads on the device.
"""
import numpy as np
import scipy.sparse as sp
import torch


class InteractionData:
    """Duck type of the reference's UIRTDataset for the MF path (weak generalisation:
    eval input == train matrix, data/dataset.py:229-248)."""

    def __init__(self, train_data, valid_target=None, test_target=None, protocol="holdout"):
        self.train_data = sp.csr_matrix(train_data)
        self.num_users, self.num_items = self.train_data.shape
        self.valid_target = valid_target
        self.test_target = test_target
        self.protocol = protocol

    @property
    def valid_input(self):
        return self.train_data

    @property
    def test_input(self):
        return self.train_data

    @classmethod
    def from_npz(cls, path):
        """load a CSR fixture written by oracle/gen_golden.py (tests/golden/ml100k_csr.npz)"""
        z = np.load(path)
        U, I = int(z["num_users"]), int(z["num_items"])

        def mk(prefix):
            ip, ix = z[prefix + "_indptr"], z[prefix + "_indices"].astype(np.int32)
            return sp.csr_matrix((np.ones(len(ix)), ix, ip), shape=(U, I))
        return cls(mk("train"), mk("valid"), mk("test"))


def csr_to_device(mat, device, row_begin=0, row_end=None):
    """scipy CSR (rows [row_begin,row_end)) -> (indptr int64, indices int32 sorted per row) on device"""
    mat = sp.csr_matrix(mat)
    if row_end is None:
        row_end = mat.shape[0]
    sub = mat[row_begin:row_end]
    sub.sort_indices()
    indptr = torch.from_numpy(sub.indptr.astype(np.int64)).to(device)
    indices = torch.from_numpy(sub.indices.astype(np.int32)).to(device)
    return indptr.contiguous(), indices.contiguous()


def synthetic_csr(num_users, num_items, degree, device, seed=2020, popularity="zipf"):
    """Synthetic positives built ON the device: every user has `degree`
    distinct items; item popularity p(i) ~ 1/(i+1) ("zipf", alpha=1) or uniform.
    Items are drawn by inverse-CDF, de-duplicated per user by re-drawing collisions
    uniformly, and sorted per row.  Returns (indptr int64 [U+1], indices int32 [U*degree]).
    """
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if popularity == "zipf":
        # inverse CDF of p(i) ~ 1/(i+1):  i = floor((I+1)^r) - 1,  r ~ U[0,1)
        r = torch.rand((num_users, degree), device=device, generator=gen, dtype=torch.float64)
        items = (torch.exp(r * float(np.log(num_items + 1.0))).floor().to(torch.int64) - 1).clamp_(0, num_items - 1)
    elif popularity == "uniform":
        items = torch.randint(0, num_items, (num_users, degree), device=device, generator=gen)
    else:
        raise ValueError(popularity)
    for _ in range(64):   # replace within-row duplicates by uniform redraws until none remain
        items, _ = items.sort(dim=1)
        dup = torch.zeros_like(items, dtype=torch.bool)
        dup[:, 1:] = items[:, 1:] == items[:, :-1]
        n = int(dup.sum())
        if n == 0:
            break
        items[dup] = torch.randint(0, num_items, (n,), device=device, generator=gen)
    else:
        raise RuntimeError("could not de-duplicate synthetic rows (degree too close to num_items?)")
    indptr = torch.arange(num_users + 1, device=device, dtype=torch.int64) * degree
    return indptr.contiguous(), items.reshape(-1).to(torch.int32).contiguous()


def load_uirt(path, separator="\t", min_item_per_user=0, min_user_per_item=0, valid_ratio=0.1,
              test_ratio=0.2, split_random=True, seed=None):
    """Read a `user item rating timestamp` text file and split it like the reference's
    UIRTDataset(protocol='holdout', generalization='weak') does:

      data/dataset.py:124-167  filter users with < min_item_per_user items, then items with
                               < min_user_per_item users; new ids in ascending raw-id order
      data/preprocess.py:12-19 weak split: FIRST a "test" part of `valid_ratio` (sic: the two
                               ratios are crossed in the reference, quirk Q8), THEN a "valid"
                               part of `test_ratio` of what is left
      data/preprocess.py:52-90 per user (ascending id): sort by timestamp, hold out
                               ceil(ratio * n) interactions chosen with np.random.choice
                               (split_random) or the last ones
    Ratings are binarised to 1 (implicit=True, data/dataset.py:46-51).  With `seed` the numpy
    global RNG is seeded first (main.py:30).  On ml-100k this reproduces the reference's
    per-user train/valid/test SIZES exactly (tests/test_loader.py against the fixture the
    reference's own loader produced); which interactions are drawn is not bit-identical (pandas'
    sort of tied timestamps), and SURVEY row f4 does not ask for that.  No on-disk cache is
    written.  Returns an InteractionData.
    """
    raw = np.loadtxt(path, delimiter=separator, dtype=np.float64, ndmin=2)
    users, items, ts = raw[:, 0].astype(np.int64), raw[:, 1].astype(np.int64), raw[:, 3]
    # filter users, then items (dataset.py:131-146)
    uid, ucnt = np.unique(users, return_counts=True)
    keep = np.isin(users, uid[ucnt >= min_item_per_user])
    users, items, ts = users[keep], items[keep], ts[keep]
    iid, icnt = np.unique(items, return_counts=True)
    keep = np.isin(items, iid[icnt >= min_user_per_item])
    users, items, ts = users[keep], items[keep], ts[keep]
    uid = np.unique(users)                       # ascending raw ids -> 0..U-1 (dataset.py:153-167)
    iid = np.unique(items)
    users = np.searchsorted(uid, users)
    items = np.searchsorted(iid, items)
    U, I = len(uid), len(iid)
    if seed is not None:
        np.random.seed(seed)

    def split(us, its, tss, ratio):
        """data/preprocess.py:52-90 on arrays; returns (kept, held-out) index arrays"""
        order = np.argsort(us, kind="stable")
        bounds = np.flatnonzero(np.diff(us[order])) + 1
        keep_idx, out_idx = [], []
        for grp in np.split(order, bounds):
            grp = grp[np.argsort(tss[grp], kind="stable")]          # sort_values(by='timestamp')
            n = len(grp)
            n_out = int(np.ceil(ratio * n)) if isinstance(ratio, float) else int(ratio)
            mask = np.ones(n, dtype=bool)
            if split_random:
                mask[np.random.choice(n, n_out, replace=False)] = False
            else:
                mask[n - n_out:] = False
            keep_idx.append(grp[mask]); out_idx.append(grp[~mask])
        return np.concatenate(keep_idx), np.concatenate(out_idx)

    all_idx = np.arange(len(users))
    k1, test = split(users, items, ts, valid_ratio)                 # sic (quirk Q8)
    k2, valid = split(users[k1], items[k1], ts[k1], test_ratio)
    train, valid = k1[k2], k1[valid]

    def csr(idx):
        m = sp.csr_matrix((np.ones(len(idx)), (users[idx], items[idx])), shape=(U, I))
        m.sum_duplicates()
        m.data[:] = 1.0
        return m
    del all_idx
    return InteractionData(csr(train), csr(valid), csr(test))
