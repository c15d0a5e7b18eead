"""tools/bench_lightgcn.py -- LightGCN step timing at the BASELINE config-5 shape (development aid)."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import recsys_pytorch_amd as pkg
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from tools.microbench import timeit
U, I, d, L, deg = 1_000_000, 100_000, 128, 3, 20
ip, ix = synthetic_csr(U, I, deg, "cuda", seed=2020)
R = sp.csr_matrix((np.ones(U * deg, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
t0 = time.time()
ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="syn")
m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 1, "reg": 0, "graph_dir": "g"}, "cuda")
m.getSparseGraph(R)
print(f"graph build + upload {time.time()-t0:.1f}s, nnz(A)={m.Graph.vals.numel()}, segments={m.Graph.num_segs}")
nnz, N = m.Graph.vals.numel(), U + I
t = timeit(lambda: rsx.spmm(m.Graph, m._E0, m._ta, S_acc=m._out), iters=5)
alg = nnz * (4 * d + 8) + 2 * N * d * 4
print(f"spmm: {t*1e3:.2f} ms, algorithmic {alg/1e9:.2f} GB -> {alg/t/1e12:.2f} TB/s")
B = 65536
u = torch.randperm(U, device="cuda")[:B].int(); i = torch.randint(0, I, (B,), device="cuda").int(); j = torch.randint(0, I, (B,), device="cuda").int()
t = timeit(lambda: m.train_step(u, i, j), iters=3, warm=1)
print(f"LightGCN train_step (L={L}, B={B}): {t*1e3:.1f} ms -> {B/t/1e6:.2f} M triplets/s")
