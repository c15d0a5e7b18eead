"""tools/ablate.py : the plain step kernel (bpr_step_kernel, unique users) under the dev library's write switches
(1 = no positive-item atomics, 2 = no negative-item atomics, 4 = no P store): where its time goes, at the base batch
65 536 and at 1M, uniform and Zipf positives"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import build as _b; os.environ["RSX_LIB"] = _b.build(dev=True)   # the -DRSX_ABLATE build (built here if run on the build host)
import torch
from recsys_pytorch_amd import rsx
from tools.microbench import timeit, zipf_items
U, d, I = 1_000_000, 128, 100_000
gen = torch.Generator(device="cuda"); gen.manual_seed(2020)
P = torch.randn(U, d, device="cuda") * 0.1
L = rsx.lib()
Q = torch.randn(I, d, device="cuda") * 0.1
G = torch.zeros_like(Q)
for B in (65536, 1000000):
    u = torch.randperm(U, device="cuda", generator=gen)[:B].to(torch.int32)
    j = torch.randint(0, I, (B,), device="cuda", dtype=torch.int32, generator=gen)
    for pop in ("uniform", "zipf"):
        i = torch.randint(0, I, (B,), device="cuda", dtype=torch.int32, generator=gen) if pop == "uniform" else zipf_items(B, I, gen)
        for mask, name in ((0, "full"), (1, "no pos atomics"), (2, "no neg atomics"), (3, "no atomics"), (4, "no P store"), (7, "loads + math only")):
            L.rsx_debug_set_ablation(mask)
            t = timeit(lambda: rsx.bpr_step(P, Q, G, u, i, j, 0.05, 1.0 / B, users_unique=True))
            print(f"B={B} {pop} positives, {name}: {t*1e6:.1f} us  {B/t/1e6:.0f} M/s", flush=True)
        G.zero_()
L.rsx_debug_set_ablation(0)
