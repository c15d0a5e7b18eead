"""oracle/diff_fuzz_loader.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (imports the reference from /root/reference).
Differential run of the reference's own code against this repository's restatement on random inputs (round 5; results:
profiles/r05_fuzz_campaign.txt).  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_fuzz_loader.py <first seed> <last seed>"""
import hashlib, os, random, shutil, sys
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(1, "/root/repo")
from data.dataset import UIRTDataset           # reference
from recsys_pytorch_amd.data import load_uirt
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    nu, ni = int(rng.integers(3, 120)), int(rng.integers(3, 90))
    n = int(rng.integers(10, 3000))
    pop = 1.0 / (1 + np.arange(ni)) ** float(rng.choice([0, 0.8])); pop /= pop.sum()
    u = rng.integers(0, nu, n) * int(rng.choice([1, 7])) + int(rng.choice([0, 1, 100]))
    i = rng.choice(ni, n, p=pop) * int(rng.choice([1, 3])) + int(rng.choice([0, 1, 50]))
    if seed % 3:     # unique pairs
        key = u.astype(np.int64) * 100000 + i
        _, first = np.unique(key, return_index=True)
        first.sort(); u, i = u[first], i[first]
    r = rng.integers(1, 6, len(u))
    t = rng.integers(0, 50 if seed % 2 else 10**9, len(u))          # many timestamp ties on odd seeds
    sep = ["\t", ",", "::"][seed % 3] if seed % 4 else "\t"
    mi, mu = int(rng.choice([0, 1, 2, 5])), int(rng.choice([0, 1, 2, 3]))
    vr, tr = float(rng.choice([0.1, 0.2, 0.34])), float(rng.choice([0.1, 0.2, 0.25]))
    sr = bool(seed % 2)
    proto = "leave_one_out" if seed % 5 == 0 else "holdout"
    lk = int(rng.choice([1, 2]))
    outs = []
    err = []
    for who in ("ref", "mine"):
        work = f"/tmp/rsx_diff_fuzz/{who}"
        shutil.rmtree(work, ignore_errors=True); os.makedirs(work)
        path = os.path.join(work, "d.data")
        with open(path, "w") as f:
            for a, b, c, d in zip(u, i, r, t):
                f.write(sep.join(str(x) for x in (a, b, c, d)) + "\n")
        try:
            if who == "ref":
                random.seed(7); np.random.seed(7)
                UIRTDataset(data_path=path, separator=sep, min_item_per_user=mi, min_user_per_item=mu, protocol=proto, generalization="weak",
                            valid_ratio=vr, test_ratio=tr, split_random=sr, leave_k=lk)
            else:
                load_uirt(path, separator=sep, min_item_per_user=mi, min_user_per_item=mu, valid_ratio=vr, test_ratio=tr, split_random=sr, seed=7,
                          cache_dir="cache", protocol=proto, leave_k=lk)
            err.append(None)
        except BaseException as e:      # noqa
            err.append(type(e).__name__ + ": " + str(e)[:100])
        d = {}
        root = os.path.join(work, "cache")
        if os.path.isdir(root):
            for sub in sorted(os.listdir(root)):
                for name in sorted(os.listdir(os.path.join(root, sub))):
                    d[sub + "/" + name] = hashlib.sha256(open(os.path.join(root, sub, name), "rb").read()).hexdigest()[:12]
        outs.append(d)
    same = outs[0] == outs[1] and (err[0] is None) == (err[1] is None)
    if not same:
        bad += 1
        print(f"seed {seed}: nu={nu} ni={ni} n={len(u)} sep={sep!r} minUI={mi},{mu} ratios={vr},{tr} random={sr} {proto} k={lk} err={err}")
        for k in sorted(set(outs[0]) | set(outs[1])):
            if outs[0].get(k) != outs[1].get(k): print("    ", k, outs[0].get(k), outs[1].get(k))
print("mismatches", bad)
