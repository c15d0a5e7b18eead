"""tools/exp_spmm_head.py : BEFORE building an LDS-resident head of the item table for the user <- item half of the LightGCN product
(VERDICT r04 item 6), its upper bound.  rsx_spmm_csr_sparse_rows does not fetch a neighbour row whose flag is 0 (it reads ONE shared zero
row instead: an L1 hit) -- with the flags of the H most popular item rows cleared, the product does everything the LDS version would do
except the LDS reads themselves (the results are wrong: timing only).  Arms: the plain product, all flags set (what the flag lookups
cost), then H = 256 / 1024 / 8192 head rows "free"; the two halves of the product and the whole of it.  BASELINE configs[4] shape."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import recsys_pytorch_amd as pkg
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
U, I, d, L, deg = 1_000_000, 100_000, 128, 3, 20
ip, ix = synthetic_csr(U, I, deg, "cuda", seed=2020)
cnt = torch.bincount(ix.long(), minlength=I)
R = sp.csr_matrix((np.ones(U * deg, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="syn")
m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": L, "node_dropout": 0.0, "split": False, "num_folds": 1, "reg": 0, "graph_dir": "g"}, "cuda")
g = m.getSparseGraph(R)


def half_of(g, which):
    keep = (g.seg_row < U) if which == "users" else (g.seg_row >= U)
    return types.SimpleNamespace(seg_row=g.seg_row[keep].contiguous(), seg_begin=g.seg_begin[keep].contiguous(), seg_len=g.seg_len[keep].contiguous(),
                                 num_segs=int(keep.sum()), indptr=g.indptr, indices=g.indices, vals=g.vals, n=g.n)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


order = torch.argsort(cnt, descending=True)
share = lambda H: float(cnt[order[:H]].sum()) / float(cnt.sum())
for name, gg in (("users half (gathers item rows)", half_of(g, "users")), ("items half (gathers user rows)", half_of(g, "items")), ("whole product", g)):
    base = timed(lambda: rsx.spmm(gg, m._E0, m._ta, S_acc=m._out))
    flags = torch.ones(U + I, dtype=torch.uint8, device="cuda")
    allset = timed(lambda: rsx.spmm(gg, m._E0, m._ta, S_acc=m._out, x_nonzero=flags))
    line = f"{name:34s} plain {base:6.3f} ms | flags all set {allset:6.3f} ms"
    for H in (256, 1024, 8192):
        f = flags.clone()
        f[U + order[:H]] = 0
        line += f" | head {H} free ({share(H)*100:4.1f} % of the gathers) {timed(lambda: rsx.spmm(gg, m._E0, m._ta, S_acc=m._out, x_nonzero=f)):6.3f} ms"
    print(line, flush=True)
