import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import build as _b; os.environ["RSX_LIB"] = _b.build(dev=True)   # the -DRSX_ABLATE build (built here if run on the build host)
import torch
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
from tools.microbench import timeit
U, I, d = 1_000_000, 100_000, 128
P = torch.randn(U, d, device="cuda") * 0.1
Q = torch.randn(I, d, device="cuda") * 0.1
ip, ix = synthetic_csr(U, I, 20, "cuda", seed=2020)
L = rsx.lib()
B, c = 1_000_000, 8
eng = BPREngine(P, Q, 0.05); eng.neg_block = c
eng.set_hot_items(torch.bincount(ix.long(), minlength=I), int(os.environ.get("HOT", "256")), int(os.environ.get("REP", "16")))
u, i, j = eng.sample(ip, ix, B)
key = eng.last_neg_key
for mask, name in ((0, "full"), (32, "Q[i] always row 0"), (64, "Q[j] always row 1"), (96, "both"), (1, "no pos flush"), (2, "no neg (LDS) adds"), (4, "no P store"), (7, "loads only")):
    L.rsx_debug_set_ablation(mask)
    t = timeit(lambda: rsx.bpr_step(P, Q, eng.G, u, i, j, 0.05, 1.0 / B, users_unique=True, neg_block=c, neg_key=key, hot=eng.hot))
    print(f"B={B} c={c} {name}: {t*1e6:.1f}us  {B/t/1e6:.0f} M/s", flush=True)
L.rsx_debug_set_ablation(0)
t = timeit(lambda: eng.sample(ip, ix, B)); print(f"sampler sorted: {t*1e6:.1f}us")
