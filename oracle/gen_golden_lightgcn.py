"""oracle/gen_golden_lightgcn.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

G6: golden vectors for LightGCN from the imported reference
(models/LightGCN.py): normalised adjacency, propagated embeddings for L in {1,2,3}, and Adam
training steps.  Fixtures are data only.  Asserts the C oracle against the reference on the way.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_lightgcn.py [name-filter]
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
np.int = int
np.float = float

import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

from models.LightGCN import LightGCN  # noqa: E402  (reference)

import oracle  # noqa: E402

torch.set_num_threads(1)


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def run(name, R, d, L, batches, seed):
    U, I = R.shape
    gdir = f"/tmp/rsx_golden_graph_{name}"
    shutil.rmtree(gdir, ignore_errors=True)
    os.makedirs(gdir)
    ds = types.SimpleNamespace(dataname=name, num_users=U, num_items=I)
    hp = {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 100,
          "reg": 1e-4, "graph_dir": gdir}                                   # conf/LightGCN.yaml
    torch.manual_seed(seed)
    m = LightGCN(ds, hp, torch.device("cpu"))
    m.Graph = m.getSparseGraph(R)                                           # LightGCN.py:70,228-258
    A_ref = sp.load_npz(os.path.join(gdir, f"{name}_s_pre_adj_mat.npz")).tocsr().astype(np.float32)
    A_ref.sort_indices()
    A = oracle.normalized_adjacency(R)
    assert A.shape == A_ref.shape and np.array_equal(A.indptr, A_ref.indptr) and np.array_equal(A.indices, A_ref.indices)
    assert np.allclose(A.data, A_ref.data, rtol=2e-7, atol=0)
    P0 = m.user_embedding.weight.detach().numpy().copy()
    Q0 = m.item_embedding.weight.detach().numpy().copy()
    orc = oracle.LightGCNOracle(P0, Q0, A_ref, L)
    m.eval()
    with torch.no_grad():
        ou, oi = m._lightgcn_embedding(m.Graph)
    pu, pi = orc.propagate()
    assert rel_err(pu, ou.numpy()) < 2e-6 and rel_err(pi, oi.numpy()) < 2e-6, "propagation differs"
    out0_u, out0_i = ou.numpy().copy(), oi.numpy().copy()
    m.train()
    losses = []
    for (u, i, j) in batches:
        m.optimizer.zero_grad()
        loss = m.process_one_batch(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
        loss.backward()
        m.optimizer.step()
        lo = orc.step(u, i, j)
        assert abs(lo - float(loss)) < 1e-5, (lo, float(loss))
        losses.append(float(loss))
    PT = m.user_embedding.weight.detach().numpy().copy()
    QT = m.item_embedding.weight.detach().numpy().copy()
    eP, eQ = rel_err(orc.P, PT), rel_err(orc.Q, QT)
    print(f"{name}: U={U} I={I} d={d} L={L} T={len(batches)} loss {losses[0]:.6f}->{losses[-1]:.6f} "
          f"oracle rel err P {eP:.2e} Q {eQ:.2e}")
    assert max(eP, eQ) < 1e-5
    cat = lambda k: np.concatenate([b[k] for b in batches]).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        P0=P0, Q0=Q0, PT=PT, QT=QT, out0_u=out0_u, out0_i=out0_i,
                        A_indptr=A_ref.indptr.astype(np.int64), A_indices=A_ref.indices.astype(np.int32),
                        A_data=A_ref.data.astype(np.float32),
                        R_indptr=sp.csr_matrix(R).indptr.astype(np.int64), R_indices=sp.csr_matrix(R).indices.astype(np.int32),
                        u=cat(0), i=cat(1), j=cat(2), batch_len=np.array([len(b[0]) for b in batches], np.int32),
                        loss=np.array(losses), num_layers=np.int32(L), lr=np.float32(1e-3))
    return m


def main():
    oracle.build()
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    global run
    run_all = run
    run = lambda name, *a: run_all(name, *a) if only in name else None
    rng = np.random.default_rng(6)
    R = sp.random(50, 40, density=0.15, format="csr", random_state=np.random.default_rng(1))
    R.data[:] = 1.0
    mk = lambda U, I, B, T: [(rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(T)]
    for L in (1, 2, 3):
        run(f"g6_lightgcn_50x40_d32_L{L}", R, 32, L, mk(50, 40, 30, 8), 20 + L)
    c = np.load(os.path.join(OUT, "ml100k_csr.npz"))
    U, I = int(c["num_users"]), int(c["num_items"])
    Rm = sp.csr_matrix((np.ones(len(c["train_indices"])), c["train_indices"].astype(np.int32), c["train_indptr"]), shape=(U, I))
    m = run("g6_lightgcn_ml100k_d64_L2", Rm, 64, 2, mk(U, I, 256, 6), 31)       # conf/LightGCN.yaml shape
    if m is not None:
        # (round 5) the reference's evaluation of that model (main.py:62-63): its Evaluator's dictionary on the valid split, its top-10
        from evaluation.evaluator import Evaluator  # noqa: E402  (reference)
        valid = sp.csr_matrix((np.ones(len(c["valid_indices"]), np.float32), c["valid_indices"].astype(np.int32), c["valid_indptr"]), shape=(U, I))
        ks = [5, 10]
        m.eval()
        scores = {k: float(v) for k, v in Evaluator(Rm, valid, protocol="holdout", ks=ks).evaluate(m).items()}
        pred32 = m.predict(np.arange(U), Rm, 1024).astype(np.float32)
        top = oracle.ref_topk(pred32, 10)
        per_user = oracle.holdout(top, ks, valid.indptr.astype(np.int64), valid.indices.astype(np.int32), use_ref=True)
        srt = -np.sort(-pred32, axis=1)
        names = [f"{mt}@{k}" for mt in ("Prec", "Recall", "NDCG") for k in ks]
        for col, n in enumerate(names):
            assert abs(float(np.mean(per_user[:, col], dtype=np.float32)) - scores[n]) < 1e-6, n
        np.savez_compressed(os.path.join(OUT, "g4_eval_lightgcn_ml100k.npz"), names=np.array(names), scores_py=np.array([scores[n] for n in names]),
                            topk10=top, per_user=per_user, gap_10=(srt[:, 9] - srt[:, 10]).astype(np.float32),
                            score_max=np.float32(np.abs(pred32[np.isfinite(pred32)]).max()))
        print("G4 LightGCN:", scores)
    # BASELINE configs[4] model shape (d=128, 3 layers) at fixture size, Zipf-ish item degrees
    rng5 = np.random.default_rng(55)
    U5, I5 = 200, 150
    pop = 1.0 / (1.0 + np.arange(I5)); pop /= pop.sum()
    rows = [np.sort(rng5.choice(I5, 12, replace=False, p=pop)) for _ in range(U5)]
    R5 = sp.csr_matrix((np.ones(U5 * 12), np.concatenate(rows), np.arange(U5 + 1) * 12), shape=(U5, I5))
    mk5 = lambda B, T: [(rng5.integers(0, U5, B), rng5.integers(0, I5, B), rng5.integers(0, I5, B)) for _ in range(T)]
    run("g6_lightgcn_200x150_d128_L3", R5, 128, 3, mk5(128, 6), 41)
    # (round 5) an emb_dim the kernels store padded (50 -> 64 columns), isolated users and items (rows / columns without interactions),
    # four layers
    rng7 = np.random.default_rng(77)
    R7 = sp.random(120, 90, density=0.03, format="csr", random_state=np.random.default_rng(7))
    R7.data[:] = 1.0
    mk7 = lambda B, T: [(rng7.integers(0, 120, B), rng7.integers(0, 90, B), rng7.integers(0, 90, B)) for _ in range(T)]
    run("g6_lightgcn_120x90_d50_L4_isolated", R7, 50, 4, mk7(64, 6), 51)


if __name__ == "__main__":
    main()
