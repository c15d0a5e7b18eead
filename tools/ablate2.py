import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for B in (1_000_000, 300_000):
  for c in (4, 8, 16):
    eng = BPREngine(P, Q, 0.05); eng.neg_block = c
    u, i, j = eng.sample(ip, ix, B)
    key = eng.last_neg_key
    for mask, name in ((0, "full"), (1, "no pos flush"), (2, "no neg (LDS) adds"), (3, "no item grads"), (4, "no P store"), (7, "loads only")):
        L.rsx_debug_set_ablation(mask)
        t = timeit(lambda: rsx.bpr_step(P, Q, eng.G, u, i, j, 0.05, 1.0 / B, users_unique=True, neg_block=c, neg_key=key))
        print(f"B={B} c={c} {name}: {t*1e6:.1f}us  {B/t/1e6:.0f} M/s", flush=True)
    L.rsx_debug_set_ablation(0)
    t = timeit(lambda: eng.sample(ip, ix, B)); print(f"B={B} c={c} sampler sorted: {t*1e6:.1f}us")
    t = timeit(lambda: rsx.apply_item_grad(Q, eng.G, 0.0)); print(f"apply: {t*1e6:.1f}us")
