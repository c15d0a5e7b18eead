"""tools/two_pass_times.py : stand-alone time of the step kernel in one launch and as item pass + user pass
(include/rsx.h RSX_ITEMS_ONLY / RSX_USERS_ONLY) on the bench workload (development aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine

U, I, d, B = 1_000_000, 100_000, 128, 1_000_000
dev = torch.device("cuda")
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev)
eng = BPREngine(P, Q, 0.05)
eng.set_neg_block(B, 8)
eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256, 16)
u, i, j = eng.sample(ip, ix, B)
kw = dict(users_unique=True, hot=eng.hot, neg_block=8, neg_key=eng.last_neg_key)
for only in (None, "items", "users"):
    ts = []
    for rep in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rsx.bpr_step(P, Q, eng.G, u, i, j, 0.05, 1.0 / B, only=only, **kw)
        b.record()
        rsx.apply_item_grad(Q, eng.G, 0.0, hot=eng.hot)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"only={only}: {sorted(ts)[len(ts) // 2]:.1f} us")
