"""tools/spmm_prof.py : the LightGCN propagation products at the BASELINE configs[4] shape, for rocprofv3 (tools/pmc_groups.py):
six forward products Y = A_hat X (models/LightGCN.py:188-197), then -- SPMM_STEP=1 -- one training step (three forward + three
backward products on the dense gradient).  STEP_PROF_META like tools/step_prof.py.
SPMM_HALF=users|items : only the segments of the user rows (they gather from the 51 MB item half of X) or of the item rows (they
gather from the 512 MB user half): the two halves of the product, each with PMC lines of its own."""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import recsys_pytorch_amd as pkg
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
U, I, d, L, deg = 1_000_000, 100_000, 128, 3, 20
ip, ix = synthetic_csr(U, I, deg, "cuda", seed=2020)
R = sp.csr_matrix((np.ones(U * deg, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="syn")
m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 1, "reg": 0, "graph_dir": "g"}, "cuda")
g = m.getSparseGraph(R)
if os.environ.get("STEP_PROF_META"):
    import bench
    json.dump({"sources_sha": bench.sources_sha("spmm"), "key": f"lightgcn_U{U}_I{I}_d{d}_L{L}" + (f"_{os.environ['SPMM_HALF']}" if os.environ.get("SPMM_HALF") else ""), "kernel": "spmm_csr_kernel", "argv": sys.argv[1:], "env": {}},
              open(os.environ["STEP_PROF_META"], "w"))
half = os.environ.get("SPMM_HALF")
if half:
    keep = (g.seg_row < U) if half == "users" else (g.seg_row >= U)
    g0 = g                                      # (keeps the hot plan's tensors alive)
    g = types.SimpleNamespace(seg_row=g.seg_row[keep].contiguous(), seg_begin=g.seg_begin[keep].contiguous(),
                              seg_len=g.seg_len[keep].contiguous(), num_segs=int(keep.sum()), indptr=g.indptr, indices=g.indices,
                              vals=g.vals, n=g.n, hot=(g.hot if half == "items" else None))   # (the rows that go by scatter are item rows)
    print(half, "segments", g.num_segs, "non-zeros", int(g.seg_len.sum()))
for _ in range(6):
    rsx.spmm(g, m._E0, m._ta, S_acc=m._out)
if os.environ.get("SPMM_STEP") == "1":
    B = 65536
    u = torch.randperm(U, device="cuda")[:B].int(); i = torch.randint(0, I, (B,), device="cuda").int(); j = torch.randint(0, I, (B,), device="cuda").int()
    m.train_step(u, i, j)
torch.cuda.synchronize()
