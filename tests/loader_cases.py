"""Random `user item rating timestamp` files for the loader's differential test (tests/test_loader.py; the expected cache digests in
tests/golden/g10_loader_random.json were written by oracle/gen_golden_loader_random.py, which runs the REFERENCE's UIRTDataset on the
very same files in the build container).  Data only: what a case is follows from its seed."""
import numpy as np

# seeds kept as fixtures: separators of one and of several characters (ml-1m's '::'), duplicate pairs, timestamp ties, both protocols,
# leave_k 2, users emptied by the item filter (133, 192, 268), and files the reference refuses (40, 53: "No objects to concatenate")
SEEDS = [1, 2, 3, 5, 10, 14, 35, 40, 53, 65, 133, 192, 268, 333]


def case(seed):
    rng = np.random.default_rng(seed)
    nu, ni = int(rng.integers(3, 120)), int(rng.integers(3, 90))
    n = int(rng.integers(10, 3000))
    pop = 1.0 / (1 + np.arange(ni)) ** float(rng.choice([0, 0.8]))
    pop /= pop.sum()
    u = rng.integers(0, nu, n) * int(rng.choice([1, 7])) + int(rng.choice([0, 1, 100]))
    i = rng.choice(ni, n, p=pop) * int(rng.choice([1, 3])) + int(rng.choice([0, 1, 50]))
    if seed % 3:                                                       # unique (user, item) pairs; otherwise duplicates stay
        key = u.astype(np.int64) * 100000 + i
        _, first = np.unique(key, return_index=True)
        first.sort()
        u, i = u[first], i[first]
    r = rng.integers(1, 6, len(u))
    t = rng.integers(0, 50 if seed % 2 else 10**9, len(u))            # many timestamp ties on odd seeds
    sep = ["\t", ",", "::"][seed % 3] if seed % 4 else "\t"
    kw = dict(separator=sep, min_item_per_user=int(rng.choice([0, 1, 2, 5])), min_user_per_item=int(rng.choice([0, 1, 2, 3])),
              valid_ratio=float(rng.choice([0.1, 0.2, 0.34])), test_ratio=float(rng.choice([0.1, 0.2, 0.25])), split_random=bool(seed % 2),
              protocol="leave_one_out" if seed % 5 == 0 else "holdout", leave_k=int(rng.choice([1, 2])))
    text = "".join(sep.join(str(x) for x in row) + "\n" for row in zip(u, i, r, t))
    return text, kw
