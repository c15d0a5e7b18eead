"""tools/ablate_step.py : stand-alone blocked step kernel on the bench workload under the development
write/load switches of rsx_debug_set_ablation (1 pos sums, 2 neg sums, 4 P store, 32 Q[i] -> row 0,
64 Q[j] -> row 1): which part of the kernel the time is sensitive to"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import build as _b; os.environ["RSX_LIB"] = _b.build(dev=True)   # the -DRSX_ABLATE build (built here if run on the build host)
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
for mask in [int(a) for a in sys.argv[1:]] or [0, 32, 64, 96, 4, 7, 103]:
    rsx.lib().rsx_debug_set_ablation(mask)
    ts = []
    for rep in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rsx.bpr_step(P, Q, eng.G, u, i, j, 0.05, 1.0 / B, **kw)
        b.record()
        rsx.apply_item_grad(Q, eng.G, 0.0, hot=eng.hot)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"ablate={mask}: {sorted(ts)[len(ts) // 2]:.1f} us")
rsx.lib().rsx_debug_set_ablation(0)
