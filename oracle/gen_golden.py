"""oracle/gen_golden.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Imports the reference (read-only, /root/reference) and writes small golden
input/output vectors under tests/golden/.  The fixtures are data only -- no
reference source or bytecode is stored.  Re-run with

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py

Golden sets:
  G1  step parity, reference loss/backward + SGD-swapped optimizer
  G1b step parity, optimizer as shipped (dense Adam lr 1e-3, models/MF.py:30)
  G2  scoring  (models/MF.py:109-112) and -inf masking (models/MF.py:114-132)
  G3  top-k    (python/func.py:4-17 and the C++ func.h:12-31 via oracle/_ref)
  G4  Evaluator.evaluate metric dicts (evaluation/evaluator.py:26-51)
  G5  PairwiseGenerator triplets on ml-100k under np.random.seed(2020)
While generating, the C restatement (oracle/mf_oracle.c) and the torch port
(oracle/torch_port.py) are asserted against the reference outputs.
"""
import os
import shutil
import sys
import types

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, REPO)
np.int = int      # utils/stats.py:15 uses aliases removed in numpy>=1.24
np.float = float

import torch  # noqa: E402

from models.MF import MF  # noqa: E402  (reference)
from data.generators import PairwiseGenerator  # noqa: E402  (reference)
from data.dataset import UIRTDataset  # noqa: E402  (reference)
from evaluation.evaluator import Evaluator  # noqa: E402  (reference)
from evaluation.backend.python.func import predict_topk_py  # noqa: E402  (reference)
from utils.general import set_random_seed  # noqa: E402  (reference)

import oracle  # noqa: E402
from oracle.torch_port import TorchMFPort  # noqa: E402

HP = lambda d: {"hidden_dim": d, "pointwise": False, "loss_func": "ce"}  # conf/MF.yaml:1-4
torch.set_num_threads(1)  # deterministic reduction order for the fixtures


def make_ref_mf(U, I, d, P0, Q0, optimizer, lr):
    ds = types.SimpleNamespace(num_users=U, num_items=I)
    m = MF(ds, HP(d), torch.device("cpu"))
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(P0))
        m.item_embedding.weight.copy_(torch.from_numpy(Q0))
    if optimizer == "sgd":  # harness-side swap; reference files untouched
        m.optimizer = torch.optim.SGD(m.parameters(), lr=lr)
    return m


def ref_step(m, u, i, j):
    """models/MF.py:64-68 verbatim call order."""
    m.optimizer.zero_grad()
    loss = m.process_one_batch(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
    loss.backward()
    gP = m.user_embedding.weight.grad.detach().numpy().copy()
    gQ = m.item_embedding.weight.grad.detach().numpy().copy()
    m.optimizer.step()
    return float(loss), gP, gQ


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def run_case(name, U, I, d, batches, optimizer, lr, seed):
    rng = np.random.default_rng(seed)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    m = make_ref_mf(U, I, d, P0, Q0, optimizer, lr)
    orc = oracle.MFOracle(P0, Q0, optimizer=optimizer, lr=lr)
    port = TorchMFPort(P0, Q0, optimizer=optimizer, lr=lr)
    losses, g1 = [], None
    for t, (u, i, j) in enumerate(batches):
        loss, gP, gQ = ref_step(m, u, i, j)
        if t == 0:
            g1 = (gP, gQ)
            ogP, ogQ, _ = orc.grad(u, i, j)
            assert rel_err(ogP, gP) < 2e-6 and rel_err(ogQ, gQ) < 2e-6, "oracle grad != reference"
        lo = orc.step(u, i, j)
        lp = port.step(u, i, j)
        assert abs(lo - loss) < 1e-5 * max(1, abs(loss)), (lo, loss)
        assert abs(lp - loss) < 1e-5 * max(1, abs(loss)), (lp, loss)
        losses.append(loss)
    PT = m.user_embedding.weight.detach().numpy().copy()
    QT = m.item_embedding.weight.detach().numpy().copy()
    eP, eQ = rel_err(orc.P, PT), rel_err(orc.Q, QT)
    pP, pQ = rel_err(port.P, PT), rel_err(port.Q, QT)
    print(f"{name}: T={len(batches)} loss0={losses[0]:.6f} lossT={losses[-1]:.6f} "
          f"oracle rel err P {eP:.2e} Q {eQ:.2e} | port P {pP:.2e} Q {pQ:.2e}")
    assert max(eP, eQ) < 1e-5 and max(pP, pQ) < 1e-5
    lens = np.array([len(b[0]) for b in batches], dtype=np.int32)
    cat = lambda k: np.concatenate([b[k] for b in batches]).astype(np.int32)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        P0=P0, Q0=Q0, PT=PT, QT=QT, gP1=g1[0], gQ1=g1[1],
        u=cat(0), i=cat(1), j=cat(2), batch_len=lens,
        loss=np.array(losses, dtype=np.float64), lr=np.float32(lr),
        optimizer=np.array(optimizer))
    return m


def random_batches(rng, U, I, B, T):
    """uniform-random triplets WITH duplicate users and items inside a batch."""
    out = []
    for _ in range(T):
        out.append((rng.integers(0, U, B).astype(np.int64),
                    rng.integers(0, I, B).astype(np.int64),
                    rng.integers(0, I, B).astype(np.int64)))
    return out


def csr_pack(m):
    m = m.tocsr()
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32)


def g23_case(name, m, mask=None):
    """G2 / G3 for one trained reference model: predict_batch_users (models/MF.py:109-112), predict's -inf mask (:114-132), top-k of
    both backends (python/func.py:4-17, func.h:12-31 via oracle/_ref) + the K-th gap; asserts oracle == reference while generating.
    mask = None: a random 8 % CSR mask"""
    Un, In = m.num_users, m.num_items
    users = np.arange(Un, dtype=np.int64)
    with torch.no_grad():
        S = m.predict_batch_users(torch.from_numpy(users)).numpy().astype(np.float32)
    So = oracle.score(m.user_embedding.weight.detach().numpy(),
                      m.item_embedding.weight.detach().numpy(), users)
    assert rel_err(So, S) < 2e-6, "oracle score != reference"
    if mask is None:
        mrng = np.random.default_rng(7)
        import scipy.sparse as sp
        mask = sp.random(Un, In, density=0.08, format="csr", random_state=mrng)
        mask.data[:] = 1.0
    pred = m.predict(users, mask, 64)  # float64 [U x I] with -inf at mask nonzeros
    mp, mi = csr_pack(mask)
    chk = oracle.mask_seen(S.copy(), users, mp, mi)
    assert np.array_equal(np.isneginf(chk), np.isneginf(pred)), "mask positions differ"
    assert np.array_equal(chk[~np.isneginf(chk)], pred.astype(np.float32)[~np.isneginf(chk)])
    pred32 = pred.astype(np.float32)  # evaluator.py:37
    save = {"mask_indptr": mp, "mask_indices": mi.astype(np.int32)}
    rows = users if Un * In <= 200_000 else users[:: max(1, Un // 48)]
    save["score_rows"] = rows.astype(np.int32)
    save["S"] = S[rows]
    for K in (5, 10, 50):
        py = predict_topk_py(pred32, K).astype(np.int32)
        cy = oracle.ref_topk(pred32, K)
        oc = oracle.topk(pred32, K)
        srt = -np.sort(-pred32, axis=1)
        gap = (srt[:, K - 1] - srt[:, K]).astype(np.float32)
        same_set = lambda a, b: all(set(a[r]) == set(b[r]) for r in range(len(a)))
        assert same_set(py, cy) and same_set(py, oc), "top-k sets differ between backends"
        vals = lambda t: np.take_along_axis(pred32, t.astype(np.int64), 1)
        assert np.array_equal(vals(cy), vals(oc)) and np.array_equal(vals(py), vals(oc))
        save[f"topk_py_{K}"] = py
        save[f"topk_cy_{K}"] = cy
        save[f"gap_{K}"] = gap
    np.savez_compressed(os.path.join(OUT, name.replace("g1_sgd", "g23") + ".npz"), **save)
    print(f"G2/G3 {name}: rows saved {len(rows)}, min gap@50 {save['gap_50'].min():.3e}")


def main():
    os.makedirs(OUT, exist_ok=True)
    oracle.build()

    # ---------------- ml-100k dataset through the reference's own loader -----
    work = "/tmp/rsx_golden/ml-100k"
    shutil.rmtree("/tmp/rsx_golden", ignore_errors=True)
    os.makedirs(work)
    shutil.copy(os.path.join(REF, "datasets/ml-100k/u.data"), work)
    set_random_seed(2020)  # main.py:30, config.py:46
    ds = UIRTDataset(data_path=os.path.join(work, "u.data"), dataname="ml-1m", separator="\t",
                     binarize_threshold=0.0, implicit=True, min_item_per_user=10,
                     min_user_per_item=1, protocol="holdout", generalization="weak",
                     holdout_users=600, valid_ratio=0.1, test_ratio=0.2, leave_k=1,
                     split_random=True)  # config.py:6-24 defaults
    U, I = ds.num_users, ds.num_items
    tr_p, tr_i = csr_pack(ds.train_data)
    va_p, va_i = csr_pack(ds.valid_target)
    te_p, te_i = csr_pack(ds.test_target)
    print("ml-100k:", U, I, len(tr_i), len(va_i), len(te_i))
    np.savez_compressed(os.path.join(OUT, "ml100k_csr.npz"), num_users=U, num_items=I,
                        train_indptr=tr_p, train_indices=tr_i.astype(np.int16),
                        valid_indptr=va_p, valid_indices=va_i.astype(np.int16),
                        test_indptr=te_p, test_indices=te_i.astype(np.int16))

    # ---------------- G5: the reference's sampler on that CSR ----------------
    np.random.seed(2020)
    gen = PairwiseGenerator(ds.train_data, num_negatives=1, num_positives_per_user=1,
                            batch_size=256, shuffle=True, device=torch.device("cpu"))
    gu, gi, gj = gen._data
    true_pos = np.mean([gi[k] in ds.train_data[gu[k]].indices for k in range(len(gu))])
    print(f"G5: {len(gu)} triplets, unique users {len(np.unique(gu))}, "
          f"'positive' is a real positive for {true_pos:.3f} of them (quirk Q1)")
    real_batches = []
    for _ in range(5):  # 5 epochs of the reference's own batching (generators.py:206-224)
        for (bu, bp, bn) in gen:
            real_batches.append((bu.numpy().copy(), bp.numpy().copy(), bn.numpy().copy()))
    np.savez_compressed(os.path.join(OUT, "g5_pairwise_ml100k.npz"),
                        users=gu.astype(np.int32), pos=gi.astype(np.int32), neg=gj.astype(np.int32),
                        frac_true_positive=np.float64(true_pos))

    # ---------------- G1 / G1b: step parity --------------------------------
    rng = np.random.default_rng(1)
    models = {}
    models["g1_sgd_200x100_d32_b64"] = run_case(
        "g1_sgd_200x100_d32_b64", 200, 100, 32, random_batches(rng, 200, 100, 64, 20), "sgd", 0.05, 11)
    models["g1_sgd_ml100k_d32_b256"] = run_case(
        "g1_sgd_ml100k_d32_b256", U, I, 32, real_batches, "sgd", 0.05, 12)
    models["g1_sgd_500x300_d64_b257"] = run_case(
        "g1_sgd_500x300_d64_b257", 500, 300, 64, random_batches(rng, 500, 300, 257, 20), "sgd", 0.05, 13)
    models["g1_sgd_400x250_d128_b512"] = run_case(
        "g1_sgd_400x250_d128_b512", 400, 250, 128, random_batches(rng, 400, 250, 512, 20), "sgd", 0.05, 14)
    run_case("g1b_adam_200x100_d32_b64", 200, 100, 32,
             random_batches(rng, 200, 100, 64, 20), "adam", 1e-3, 15)
    run_case("g1b_adam_ml100k_d32_b256", U, I, 32, real_batches[:12], "adam", 1e-3, 16)

    # ---------------- G2 / G3: scoring, masking, top-k on the G1 end states --
    for name, m in models.items():
        g23_case(name, m, ds.train_data if (m.num_users == U and m.num_items == I) else None)

    # ---------------- G4: Evaluator.evaluate on the ml-100k end state --------
    m = models["g1_sgd_ml100k_d32_b256"]
    ev = Evaluator(ds.valid_input, ds.valid_target, protocol="holdout", ks=[5, 10])
    scores_py = {k: float(v) for k, v in ev.evaluate(m).items()}
    users = np.arange(U, dtype=np.int64)
    pred32 = m.predict(users, ds.valid_input, 1024).astype(np.float32)
    top = oracle.ref_topk(pred32, 10)
    res_ref = oracle.holdout(top, [5, 10], va_p, va_i, use_ref=True)
    res_orc = oracle.holdout(oracle.topk(pred32, 10), [5, 10], va_p, va_i, use_ref=False)
    assert np.allclose(res_ref, res_orc, atol=1e-6), "oracle holdout != reference header"
    names = [f"{mt}@{k}" for mt in ("Prec", "Recall", "NDCG") for k in (5, 10)]
    scores_cy = {n: float(np.mean(res_ref[:, c], dtype=np.float32)) for c, n in enumerate(names)}
    print("G4 python backend:", scores_py)
    print("G4 native header :", scores_cy)
    for n in names:
        assert abs(scores_py[n] - scores_cy[n]) < 1e-6
    np.savez_compressed(os.path.join(OUT, "g4_eval_ml100k.npz"),
                        names=np.array(names),
                        scores_py=np.array([scores_py[n] for n in names], dtype=np.float64),
                        scores_cy=np.array([scores_cy[n] for n in names], dtype=np.float64),
                        topk10=top, per_user=res_ref)
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"fixtures written to {OUT}: {sz / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
