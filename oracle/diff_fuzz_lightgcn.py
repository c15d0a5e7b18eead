"""oracle/diff_fuzz_lightgcn.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (imports the reference from /root/reference).
Differential run of the reference's own code against this repository's restatement on random inputs (round 5; results:
profiles/r05_fuzz_campaign.txt).  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_fuzz_lightgcn.py <first seed> <last seed>"""
import os, shutil, sys, types
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(1, "/root/repo")
np.int = int; np.float = float
import scipy.sparse as sp, torch
torch.set_num_threads(1)
from models.LightGCN import LightGCN
import oracle
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    U, I = int(rng.integers(2, 200)), int(rng.integers(2, 150))
    d, L = int(rng.integers(1, 130)), int(rng.integers(1, 5))
    dens = float(rng.choice([0.005, 0.05, 0.3]))
    R = sp.random(U, I, density=dens, format="csr", random_state=np.random.default_rng(seed + 1), dtype=np.float32)
    R.data[:] = 1.0 if seed % 2 else rng.choice([1.0, 1.0, 2.0, 3.0], R.nnz)     # (a pair listed twice in the raw file holds 2 in the train matrix)
    if R.nnz == 0:
        continue
    gdir = "/tmp/rsx_diff_fuzz/graph"; shutil.rmtree(gdir, ignore_errors=True); os.makedirs(gdir)
    ds = types.SimpleNamespace(dataname="f", num_users=U, num_items=I)
    hp = {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 100, "reg": 1e-4, "graph_dir": gdir}
    torch.manual_seed(seed)
    m = LightGCN(ds, hp, torch.device("cpu"))
    m.Graph = m.getSparseGraph(R)
    A_ref = sp.load_npz(os.path.join(gdir, "f_s_pre_adj_mat.npz")).tocsr().astype(np.float32); A_ref.sort_indices(); A_ref.eliminate_zeros()
    A = oracle.normalized_adjacency(R); A.eliminate_zeros()
    ctx = f"seed {seed}: U={U} I={I} d={d} L={L} nnz={R.nnz} isolated users {(np.diff(R.indptr) == 0).sum()} items {(np.diff(R.tocsc().indptr) == 0).sum()}"
    if not (A.shape == A_ref.shape and np.array_equal(A.indptr, A_ref.indptr) and np.array_equal(A.indices, A_ref.indices) and np.allclose(A.data, A_ref.data, rtol=2e-7, atol=0)):
        bad += 1; print(ctx, "adjacency differs"); continue
    P0 = m.user_embedding.weight.detach().numpy().copy(); Q0 = m.item_embedding.weight.detach().numpy().copy()
    orc = oracle.LightGCNOracle(P0, Q0, oracle.normalized_adjacency(R), L)
    m.eval()
    with torch.no_grad():
        ou, oi = m._lightgcn_embedding(m.Graph)
    pu, pi = orc.propagate()
    sc = max(np.abs(ou.numpy()).max(), np.abs(oi.numpy()).max())
    if max(np.abs(pu - ou.numpy()).max(), np.abs(pi - oi.numpy()).max()) > 2e-6 * sc:
        bad += 1; print(ctx, "propagation differs"); continue
    m.train()
    B = int(rng.integers(1, 300))
    for t in range(2):
        u, i, j = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
        m.optimizer.zero_grad()
        loss = m.process_one_batch(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
        loss.backward(); m.optimizer.step()
        lo = orc.step(u, i, j)
        if abs(lo - float(loss)) > 1e-5 * max(1, abs(float(loss))):
            bad += 1; print(ctx, "loss differs", t, lo, float(loss)); break
print("bad", bad)
