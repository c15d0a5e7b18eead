"""oracle/gen_golden_shipped_config.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

The reference exactly as it ships (BASELINE configs[0]): conf/MF.yaml's hidden_dim 50, the optimizer of models/MF.py:30 (Adam, lr 1e-3), the
ml-100k split of its own loader and the first 12 batches of its own PairwiseGenerator (taken from the fixture g1b_adam_ml100k_d32_b256,
which oracle/gen_golden.py recorded from that generator) -> tests/golden/g1b_adam_ml100k_d50_b256.npz, through oracle/gen_golden.py's
run_case (which asserts oracle == reference while generating).  Round 5.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_shipped_config.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)


def main():
    G.oracle.build()
    z = np.load(os.path.join(G.OUT, "g1b_adam_ml100k_d32_b256.npz"))
    U, I = z["P0"].shape[0], z["Q0"].shape[0]
    cuts = np.concatenate([[0], np.cumsum(z["batch_len"])])
    batches = [tuple(z[k][cuts[t]:cuts[t + 1]].astype(np.int64) for k in ("u", "i", "j")) for t in range(len(z["batch_len"]))]
    m = G.run_case("g1b_adam_ml100k_d50_b256", U, I, 50, batches, "adam", 1e-3, 61)
    # ... and the reference's evaluation of that model (main.py:62-63, evaluation/evaluator.py:26-54): its Evaluator on the valid split
    import scipy.sparse as sp
    c = np.load(os.path.join(G.OUT, "ml100k_csr.npz"))
    csr = lambda part: sp.csr_matrix((np.ones(len(c[part + "_indices"]), np.float32), c[part + "_indices"].astype(np.int32), c[part + "_indptr"]),
                                     shape=(U, I))
    train, valid = csr("train"), csr("valid")
    ks = [5, 10]
    scores = {k: float(v) for k, v in G.Evaluator(train, valid, protocol="holdout", ks=ks).evaluate(m).items()}
    users = np.arange(U, dtype=np.int64)
    pred32 = m.predict(users, train, 1024).astype(np.float32)
    top = G.oracle.ref_topk(pred32, 10)
    per_user = G.oracle.holdout(top, ks, valid.indptr.astype(np.int64), valid.indices.astype(np.int32), use_ref=True)
    srt = -np.sort(-pred32, axis=1)
    names = [f"{mt}@{k}" for mt in ("Prec", "Recall", "NDCG") for k in ks]
    for col, n in enumerate(names):
        assert abs(float(np.mean(per_user[:, col], dtype=np.float32)) - scores[n]) < 1e-6, n
    np.savez_compressed(os.path.join(G.OUT, "g4_eval_ml100k_d50.npz"), names=np.array(names), scores_py=np.array([scores[n] for n in names]),
                        topk10=top, per_user=per_user, gap_10=(srt[:, 9] - srt[:, 10]).astype(np.float32))
    print("G4 at the shipped configuration:", scores)


if __name__ == "__main__":
    main()
