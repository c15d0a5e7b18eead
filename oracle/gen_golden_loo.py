"""oracle/gen_golden_loo.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

G9: the leave-one-out protocol of the reference.
 * its UIRTDataset(protocol='leave_one_out', leave_k=1) on a /tmp copy of ml-100k (data/dataset.py:170-179): cache
   directory name, sizes and sha256 digests of the files it writes (the build's loader must write the same bytes), and
   the train / valid / test matrices as CSR;
 * its Evaluator(protocol='leave_one_out') (evaluation/evaluator.py:10-54 -> backend/python/loo.py:11-32) on a seeded
   model: the rankings it evaluates, the per-user HR / NDCG it accumulates and the means it returns.
Asserts oracle == reference (and, when built, the reference's loo.h through oracle/_ref) while generating.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_loo.py
"""
import hashlib
import json
import os
import random
import shutil
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)
from data.dataset import UIRTDataset  # noqa: E402  (reference)
from evaluation.evaluator import Evaluator  # noqa: E402  (reference)


def main():
    G.oracle.build(with_ref=True)
    work = "/tmp/rsx_golden_loo/ml-100k"
    shutil.rmtree("/tmp/rsx_golden_loo", ignore_errors=True)
    os.makedirs(work)
    shutil.copy(os.path.join(G.REF, "datasets/ml-100k/u.data"), work)
    random.seed(2020); np.random.seed(2020)          # utils/general.py:31-38 via main.py:30
    ds = UIRTDataset(data_path=os.path.join(work, "u.data"), separator="\t", min_item_per_user=10, min_user_per_item=1,
                     protocol="leave_one_out", generalization="weak", leave_k=1, split_random=True)
    cache_root = os.path.join(work, "cache")
    (sub,) = os.listdir(cache_root)
    meta = {"cache_subdir": sub, "files": {}}
    for name in sorted(os.listdir(os.path.join(cache_root, sub))):
        raw = open(os.path.join(cache_root, sub, name), "rb").read()
        meta["files"][name] = {"bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest(), "head": raw.decode().splitlines()[:3]}
    json.dump(meta, open(os.path.join(G.OUT, "g9_ml100k_loo_cache.json"), "w"), indent=1, sort_keys=True)
    U, I = ds.num_users, ds.num_items
    tr_p, tr_i = G.csr_pack(ds.train_data)
    va_p, va_i = G.csr_pack(ds.valid_target)
    te_p, te_i = G.csr_pack(ds.test_target)
    print("ml-100k leave-one-out:", U, I, len(tr_i), len(va_i), len(te_i), sub)

    # a seeded model state, evaluated by the reference's own Evaluator on the validation targets
    rng = np.random.default_rng(9)
    d, ks = 32, [1, 5, 10]
    P0 = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
    # make the held-out item rank high for a third of the users, so that hits exist at every K
    tgt = va_i[va_p[:-1]]
    for u in range(0, U, 3):
        Q0[tgt[u]] += 0.5 * P0[u]
    m = G.make_ref_mf(U, I, d, P0, Q0, "adam", 1e-3)
    ev = Evaluator(ds.valid_input, ds.valid_target, "leave_one_out", ks)
    scores = ev.evaluate(m)
    hist = ev.evaluate(m, mean=False)
    output = m.predict(np.arange(U), ds.valid_input, 1024)
    pred = ev.predict_topk(output.astype(np.float32), max(ks))
    per_user = np.stack([np.asarray(hist["%s@%d" % (mt, k)], dtype=np.float32) for mt in ("HR", "NDCG") for k in ks], axis=1)
    got = G.oracle.loo(pred, ks, tgt)
    assert np.allclose(got, per_user, atol=1e-6), "oracle loo != reference python backend"
    ref = G.oracle.loo(pred, ks, tgt, use_ref=True)
    assert np.allclose(ref, per_user, atol=1e-6), "reference loo.h != reference python backend"
    print({k: float(v) for k, v in scores.items()})
    np.savez_compressed(os.path.join(G.OUT, "g9_loo_eval_ml100k.npz"), num_users=U, num_items=I,
                        train_indptr=tr_p, train_indices=tr_i.astype(np.int16), valid_indptr=va_p, valid_indices=va_i.astype(np.int16),
                        test_indptr=te_p, test_indices=te_i.astype(np.int16), P0=P0, Q0=Q0, ks=np.array(ks, np.int32),
                        topk10=pred.astype(np.int32), per_user=per_user,
                        score_names=np.array(list(scores.keys())), score_values=np.array([scores[k] for k in scores], np.float64))


if __name__ == "__main__":
    main()
