"""oracle/diff_fuzz_topk.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (imports the reference from /root/reference).
Differential run of the reference's own code against this repository's restatement on random inputs (round 5; results:
profiles/r05_fuzz_campaign.txt).  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_fuzz_topk.py <first seed> <last seed>"""
import sys
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(1, "/root/repo")
from evaluation.backend.python.func import predict_topk_py
import oracle
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    rng = np.random.default_rng(seed)
    rows, I = int(rng.integers(1, 60)), int(rng.integers(2, 3000))
    K = int(rng.integers(1, min(I - 1, 100) + 1))
    S = rng.standard_normal((rows, I)).astype(np.float32)
    if seed % 3 == 0:
        S[:, rng.integers(0, I, I // 2)] = -np.inf         # seen items
    ref = predict_topk_py(S.astype(np.float64) if seed % 2 else S, K)
    got = oracle.topk(S, K)
    rt = oracle.ref_topk(S, K) if oracle.ref_lib() is not None else got
    v_ref = np.take_along_axis(S, ref, 1); v_got = np.take_along_axis(S, got.astype(np.int64), 1); v_rt = np.take_along_axis(S, rt.astype(np.int64), 1)
    if not (np.array_equal(v_ref, v_got) and np.array_equal(v_rt, v_got)):
        bad += 1; print("values differ", seed, rows, I, K)
    uniq = np.array([len(np.unique(r[np.isfinite(r)])) == np.isfinite(r).sum() and np.isfinite(r).sum() >= K for r in S])
    if not (np.array_equal(ref[uniq], got[uniq]) and np.array_equal(rt[uniq], got[uniq])):
        bad += 1; print("indices differ", seed, rows, I, K)
print("bad", bad)
