"""Interaction data for the BPR-MF path: CSR holders and the synthetic workload.

What the hot path needs from the reference's dataset code is only `.num_users`,
`.num_items`, `.train_data` (scipy CSR) and the eval input/target matrices
(models/MF.py:16-17,45; main.py:62-63).  `InteractionData` is that duck type;
`load_uirt` restates the reference's filter / remap / weak holdout split and its
on-disk cache format (data/dataset.py:92-218,
data/preprocess.py:12-90); `synthetic_csr` builds the BASELINE.json workloads on
the device.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch


class InteractionData:
    """Duck type of the reference's UIRTDataset for the MF path (weak generalisation:
    eval input == train matrix, data/dataset.py:229-248)."""

    def __init__(self, train_data, valid_target=None, test_target=None, protocol="holdout", dataname=None, user2id=None, item2id=None):
        self.train_data = sp.csr_matrix(train_data)
        self.num_users, self.num_items = self.train_data.shape
        self.valid_target = valid_target
        self.test_target = test_target
        self.protocol = protocol
        # what else a caller of the reference's UIRTDataset reads (data/dataset.py:19,55-56,67): the name LightGCN files its graph
        # under, the raw-id -> id maps, the users that have a train row (weak generalisation: the same list three times)
        self.dataname = dataname
        self.user2id, self.item2id = user2id, item2id
        users = np.flatnonzero(np.diff(self.train_data.indptr)).tolist()
        self.train_users = self.valid_users = self.test_users = users

    @property
    def valid_input(self):
        return self.train_data

    @property
    def test_input(self):
        """weak generalisation: the test-time input is train + valid (data/dataset.py:236-241)"""
        return self.train_data if self.valid_target is None else self.train_data + self.valid_target

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


def _cache_subdir(valid_ratio, test_ratio, split_random, min_item_per_user, min_user_per_item, seed,
                  protocol="holdout", leave_k=1):
    """data/dataset.py:210-220 (weak generalisation): the directory name keys every split parameter"""
    how = "random" if split_random else "time"
    if protocol == "leave_one_out":
        return "loo_%d_weak_%s_minUI_%d_%d_seed%d" % (leave_k, how, min_item_per_user, min_user_per_item, seed)
    return "holdout_%.2f_%.2f_weak_%s_minUI_%d_%d_seed%d" % (valid_ratio, test_ratio, how, min_item_per_user,
                                                             min_user_per_item, seed)


CACHE_FILES = ("train.csv", "valid.csv", "test.csv", "user_map", "item_map")


def _write_cache(cdir, parts, user_raw, item_raw):
    """the reference's on-disk cache (data/dataset.py:176-181,214-218): `<part>.csv` rows
    `user,item,rating,timestamp` with NEW ids, float rating / timestamp as pandas prints them
    (DataFrame.to_csv(index=False, header=False)), and `user_map` / `item_map` lines "raw, new"."""
    os.makedirs(cdir, exist_ok=True)
    for name, (u, it, r, t) in parts.items():
        with open(os.path.join(cdir, name + ".csv"), "wt") as f:
            f.write("".join("%d,%d,%r,%r\n" % (a, b, float(c), float(d)) for a, b, c, d in zip(u, it, r, t)))
    for name, raw in (("user_map", user_raw), ("item_map", item_raw)):
        with open(os.path.join(cdir, name), "wt") as f:
            f.write("".join("%d, %d\n" % (int(old), new) for new, old in enumerate(raw)))


def _read_cache(cdir, protocol="holdout"):
    """data/dataset.py:43-66,209-212: id maps give the table sizes, the three csv files the matrices"""
    maps = []
    for name in ("user_map", "item_map"):
        with open(os.path.join(cdir, name), "rt") as f:              # dataset.py:229-235
            maps.append({int(a): int(b) for a, b in (line.strip().split(", ") for line in f if line.strip())})
    U, I = len(maps[0]), len(maps[1])
    mats = []
    for name in ("train", "valid", "test"):
        raw = np.loadtxt(os.path.join(cdir, name + ".csv"), delimiter=",", dtype=np.float64, ndmin=2)
        m = sp.csr_matrix((np.ones(len(raw)), (raw[:, 0].astype(np.int64), raw[:, 1].astype(np.int64))), shape=(U, I))
        m.sum_duplicates()                                      # implicit=True: every rating becomes 1 at load (dataset.py:46-51); a pair
        #                                                         listed twice in the file then holds 2, as in the reference's
        #                                                         csr_matrix((ones, (users, items))) (utils/types.py:5-11)
        mats.append(m)
    return InteractionData(*mats, protocol=protocol, dataname=os.path.basename(os.path.dirname(os.path.dirname(cdir))), user2id=maps[0], item2id=maps[1])


def load_uirt(path, separator="\t", min_item_per_user=0, min_user_per_item=0, valid_ratio=0.1,
              test_ratio=0.2, split_random=True, seed=None, cache_dir=None, cache_seed=1234,
              protocol="holdout", leave_k=1):
    """Read a `user item rating timestamp` text file and split it like the reference's
    UIRTDataset(protocol='holdout', generalization='weak') does:

      data/dataset.py:124-167  filter users with < min_item_per_user items, then items with
                               < min_user_per_item users; new ids in ascending raw-id order
      data/preprocess.py:12-19 weak split: FIRST a "test" part of `valid_ratio` (sic: the two
                               ratios are crossed in the reference, quirk Q8), THEN a "valid"
                               part of `test_ratio` of what is left
      data/preprocess.py:52-90 per user (ascending id): sort by timestamp (pandas' default
                               quicksort: ties in numpy's introsort order), hold out ceil(ratio * n)
                               interactions chosen with np.random.choice (split_random) or the last
      data/dataset.py:92-122,176-218  on-disk cache: with `cache_dir` (the reference's default is
                               'cache') the split is written to / read back from
                               <dirname(path)>/<cache_dir>/holdout_<v>_<t>_weak_<random|time>_minUI_<a>_<b>_seed<cache_seed>/
                               {train,valid,test}.csv + user_map + item_map, byte for byte the
                               files the reference writes (tests/test_loader.py against digests
                               recorded from the reference's own cache).  `cache_seed` only names
                               the directory, as in the reference (its `seed` argument, default 1234).
    protocol='leave_one_out' (data/dataset.py:170-179): the same two splits with COUNTS instead of ratios -- `leave_k`
    interactions per user held out for "test", then `leave_k` of the rest for "valid" (preprocess.py:60-63); cache
    directory `loo_<k>_weak_<random|time>_minUI_<a>_<b>_seed<cache_seed>` (dataset.py:214-215).
    Ratings are binarised to 1 (implicit=True, data/dataset.py:46-51).  With `seed` the numpy
    global RNG is seeded first (main.py:30).  On ml-100k with seed 2020 this reproduces the
    reference's train/valid/test matrices exactly.  Returns an InteractionData.
    """
    cdir = None
    if cache_dir is not None:
        cdir = os.path.join(os.path.dirname(os.path.abspath(path)), cache_dir,
                            _cache_subdir(valid_ratio, test_ratio, split_random, min_item_per_user, min_user_per_item, cache_seed,
                                          protocol, leave_k))
        if all(os.path.exists(os.path.join(cdir, f)) for f in CACHE_FILES):     # dataset.py:183-191
            return _read_cache(cdir, protocol)
    if len(separator) == 1:
        raw = np.loadtxt(path, delimiter=separator, dtype=np.float64, ndmin=2)
    else:
        # a separator of several characters -- ml-1m's '::' -- is a regular expression to the reference's reader (dataset.py:115-118:
        # pandas, engine='python'); numpy's reader takes single characters only
        import re
        cut = re.compile(separator)
        with open(path) as f:
            raw = np.array([[float(x) for x in cut.split(line.rstrip("\r\n"))[:4]] for line in f if line.strip()], dtype=np.float64).reshape(-1, 4)
    users, items, ratings, ts = raw[:, 0].astype(np.int64), raw[:, 1].astype(np.int64), raw[:, 2], raw[:, 3]
    # filter users, then items (dataset.py:131-146)
    uid, ucnt = np.unique(users, return_counts=True)
    keep = np.isin(users, uid[ucnt >= min_item_per_user])
    users, items, ratings, ts = users[keep], items[keep], ratings[keep], ts[keep]
    # the reference numbers the users that passed the USER filter (dataset.py:135-137,153-157: its per-user counts are taken before
    # the item filter): a user whose every item falls to the item filter keeps an id and an empty row.  (Round 5: found by running
    # the reference's loader and this one on 300 random files; the ids used to be taken after both filters)
    uid = np.unique(users)                       # ascending raw ids -> 0..U-1 (dataset.py:153-167)
    iid, icnt = np.unique(items, return_counts=True)
    keep = np.isin(items, iid[icnt >= min_user_per_item])
    users, items, ratings, ts = users[keep], items[keep], ratings[keep], ts[keep]
    iid = np.unique(items)
    if len(users) == 0:      # (the reference fails here too: pandas' "No objects to concatenate" from the split of an empty frame)
        raise ValueError("No objects to concatenate: no interaction is left after the min_item_per_user / min_user_per_item filters")
    users = np.searchsorted(uid, users)
    items = np.searchsorted(iid, items)
    U, I = len(uid), len(iid)
    if seed is not None:
        np.random.seed(seed)

    def split(idx, ratio):
        """data/preprocess.py:52-90 on the rows `idx` (in frame order); returns (kept, held-out)
        row ids, each in the order the reference concatenates them"""
        us = users[idx]
        order = np.argsort(us, kind="stable")                       # groupby('user'): ascending, rows in frame order
        bounds = np.flatnonzero(np.diff(us[order])) + 1
        keep_idx, out_idx = [], []
        for grp in np.split(idx[order], bounds):
            grp = grp[np.argsort(ts[grp], kind="quicksort")]        # sort_values(by='timestamp'), pandas' default kind
            n = len(grp)
            n_out = int(np.ceil(ratio * n)) if isinstance(ratio, float) else int(ratio)
            mask = np.ones(n, dtype=bool)
            if split_random:
                mask[np.random.choice(n, n_out, replace=False)] = False
            else:
                mask[n - n_out:] = False
            keep_idx.append(grp[mask]); out_idx.append(grp[~mask])
        keep_all, out_all = np.concatenate(keep_idx), np.concatenate(out_idx)
        if len(keep_all) == 0 or len(out_all) == 0:                 # preprocess.py:87-88: pd.concat of an empty list
            raise ValueError("No objects to concatenate: a split left no interaction on one of its sides")
        return keep_all, out_all

    if protocol not in ("holdout", "leave_one_out"):
        raise ValueError(f"{protocol} is not a valid protocol.")                 # dataset.py:189-190
    first, second = (int(leave_k), int(leave_k)) if protocol == "leave_one_out" else (float(valid_ratio), float(test_ratio))
    k1, test = split(np.arange(len(users)), first)                  # sic (quirk Q8)
    train, valid = split(k1, second)

    def csr(idx):
        m = sp.csr_matrix((np.ones(len(idx)), (users[idx], items[idx])), shape=(U, I))
        m.sum_duplicates()                                      # (a pair listed twice holds 2, as in the reference: see _read_cache)
        return m
    if cdir is not None:
        _write_cache(cdir, {n: (users[ix], items[ix], ratings[ix], ts[ix]) for n, ix in
                            (("train", train), ("valid", valid), ("test", test))}, uid, iid)
    return InteractionData(csr(train), csr(valid), csr(test), protocol=protocol, dataname=os.path.basename(os.path.dirname(os.path.abspath(path))),
                           user2id={int(old): new for new, old in enumerate(uid)}, item2id={int(old): new for new, old in enumerate(iid)})
