"""oracle/diff_fuzz_evaluator.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (imports the reference from /root/reference).
The reference's Evaluator (evaluation/evaluator.py:10-54, python backend) against recsys_pytorch_amd.evaluator.Evaluator on random
score matrices, both protocols, any cut-offs: the same score dictionaries.  (The package's Evaluator gets a stub model whose predict_topk
is the oracle's partial sort of the same scores: no GPU needed; its metric loops are the host functions of librsx.so.)
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_fuzz_evaluator.py <first seed> <last seed>"""
import sys
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(1, "/root/repo")
np.int = int; np.float = float
import scipy.sparse as sp
from evaluation.evaluator import Evaluator as RefEvaluator       # reference
from recsys_pytorch_amd.evaluator import Evaluator
import oracle


class RefStub:
    def __init__(self, S): self.S = S
    def eval(self): pass
    def predict(self, eval_users, eval_pos, test_batch_size):
        out = self.S.astype(np.float64).copy()
        out[eval_pos.nonzero()] = float("-inf")
        return out


class MyStub(RefStub):
    def predict_topk(self, eval_users, eval_pos, K, test_batch_size=1024, want_values=False):
        return oracle.topk(self.predict(eval_users, eval_pos, test_batch_size).astype(np.float32), K)


bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    rng = np.random.default_rng(seed)
    U, I = int(rng.integers(1, 150)), int(rng.integers(12, 300))
    if seed % 25 == 7:                       # enough users for the metric loops to run over host threads (csrc/rsx_eval.hip: from 8 192 on)
        U, I = int(rng.integers(8_192, 20_000)), int(rng.integers(40, 90))
    S = rng.standard_normal((U, I)).astype(np.float32)
    loo = seed % 3 == 0
    seen = sp.random(U, I, density=0.05, format="csr", random_state=np.random.default_rng(seed + 5), dtype=np.float32)
    tgt = sp.lil_matrix((U, I), dtype=np.float32)
    for u in range(U):
        free = np.setdiff1d(np.arange(I), seen[u].indices)
        n = 1 if loo else int(rng.integers(1, 6))
        tgt[u, rng.choice(free, min(n, len(free)), replace=False)] = 1.0
    tgt = tgt.tocsr()
    ks = sorted({int(k) for k in rng.integers(1, 11, int(rng.integers(1, 4)))})
    proto = "leave_one_out" if loo else "holdout"
    a = RefEvaluator(seen, tgt, proto, ks).evaluate(RefStub(S))
    b = Evaluator(seen, tgt, proto, ks).evaluate(MyStub(S))
    if set(a) != set(b) or any(abs(float(a[k]) - float(b[k])) > 1e-6 for k in a):
        bad += 1
        print("seed", seed, proto, ks, {k: (float(a[k]), float(b.get(k, np.nan))) for k in a})
print("bad", bad)
