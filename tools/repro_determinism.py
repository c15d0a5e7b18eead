"""does the same engine set-up give the same relabelling / sampler tables / first batch in two processes?  (tools/repro_one_run.py)"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from recsys_pytorch_amd.sharded import BPREngine
from recsys_pytorch_amd import rsx
U, I, d, B, deg, chunks = 2364, int(sys.argv[1]) if len(sys.argv) > 1 else 782, 64, 1037, 7, 3
hot_on = (sys.argv[2] if len(sys.argv) > 2 else "hot") == "hot"
h = lambda t: hashlib.md5(np.ascontiguousarray(t.cpu().numpy() if torch.is_tensor(t) else t).tobytes()).hexdigest()[:10]
rng = np.random.default_rng(500)
p = 1.0 / np.arange(1, I + 1) ** 0.7
p /= p.sum()
rows = [np.sort(rng.choice(I, deg, replace=False, p=p)) for _ in range(U)]
dev = torch.device("cuda", 0)
indptr = (torch.arange(U + 1, dtype=torch.int64) * deg).to(dev)
indices = torch.from_numpy(np.concatenate(rows).astype(np.int32)).to(dev)
P = (torch.randn(U, d, generator=torch.Generator().manual_seed(100)) * 0.1).to(dev)
Q = (torch.randn(I, d, generator=torch.Generator().manual_seed(7)) * 0.1).to(dev)
eng = BPREngine(P, Q, 0.05 * B, seed=11)
eng.set_neg_block(B, 8)
if B < 2 * I:
    eng.sorted_min_batch = 1
if hot_on:
    eng.set_hot_items(torch.bincount(indices.long(), minlength=I), 32, 4)
eng.set_chunks(chunks)
tr = eng.native_trainer(indptr, indices, B)
r = eng._relabel
tr.run(1)
torch.cuda.synchronize()
u, i, j = tr.last_batch()[:3]
print("I", I, "hot", hot_on, "rank_item", h(r["rank_item"]), "indices", h(r["indices"]), "cdf", h(r["cdf"]) if r["cdf"] is not None else None,
      "hot items", h(torch.sort(r["hot"].items).values) if r["hot"] is not None else None, "u", h(u), "i", h(i), "j", h(j), "cp", h(tr.last_chunk_pos()),
      "P", h(P))
deg_ = (indptr[1:] - indptr[:-1]).double()
w = torch.repeat_interleave(1.0 / deg_.clamp_min(1.0), indptr[1:] - indptr[:-1])
mass = torch.zeros(I, dtype=torch.float64, device=dev).index_add_(0, indices.long(), w)
m2 = np.bincount(indices.cpu().numpy(), weights=w.cpu().numpy(), minlength=I)
print("   mass (device index_add_)", h(mass), "mass (numpy bincount)", h(m2), "max diff", float(np.abs(mass.cpu().numpy() - m2).max()),
      "distinct w", len(np.unique(w.cpu().numpy())))
from recsys_pytorch_amd.sharded import deal_items_to_ranges
base, rem = divmod(I, chunks)
cap = np.array([base + (k < rem) for k in range(chunks)], dtype=np.int64)
a1 = deal_items_to_ranges(mass.cpu().numpy(), cap, np.random.default_rng(5))
a2 = deal_items_to_ranges(m2, cap, np.random.default_rng(5))
print("   assign from device mass", h(a1), "from numpy mass", h(a2))
