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
