"""tools/lightgcn_fit_time.py : LightGCN.fit at BASELINE configs[4]'s shape (1M x 100K, d = 128, 3 layers, batch 65 536) through the model class:
the first fit (graph build + set-up), further fits on the same matrix, ms per batch against bench.py's train_step figure, one evaluation."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import recsys_pytorch_amd as pkg
from recsys_pytorch_amd.data import synthetic_csr
U, I, d, L, B = 1_000_000, 100_000, 128, 3, 65536
ip, ix = synthetic_csr(U, I, 20, "cuda", seed=2020)
R = sp.csr_matrix((np.ones(U * 20, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
rng = np.random.default_rng(1)
T = sp.csr_matrix((np.ones(U * 5, np.float32), rng.integers(0, I, U * 5), np.arange(U + 1) * 5), shape=(U, I))
ds = pkg.InteractionData(R, T, T, dataname="syn")
m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 1, "reg": 0, "graph_dir": "g", "lr": 1e-3}, "cuda")
cfg = lambda n: types.SimpleNamespace(batch_size=B, num_epochs=n, verbose=0, test_from=10**9, test_step=1)
nb = -(-U // B)
for name, n in (("first fit, 1 epoch (graph build + set-up)", 1), ("second fit, 1 epoch", 1), ("third fit, 3 epochs", 3)):
    torch.cuda.synchronize(); t = time.perf_counter(); m.fit(ds, cfg(n)); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name}: {dt * 1e3:.1f} ms = {dt / (n * nb) * 1e3:.2f} ms per batch of {B} ({nb} batches per epoch)")
ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "holdout", [10, 50])
ev.evaluate(m); torch.cuda.synchronize()
t = time.perf_counter(); ev.evaluate(m); torch.cuda.synchronize(); print(f"evaluate: {(time.perf_counter() - t) * 1e3:.1f} ms")
