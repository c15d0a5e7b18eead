import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import build as _b; os.environ["RSX_LIB"] = _b.build(dev=True)   # the -DRSX_ABLATE build (built here if run on the build host)
import torch
from recsys_pytorch_amd import rsx
from tools.microbench import timeit, zipf_items
U, d = 1_000_000, 128
gen = torch.Generator(device="cuda"); gen.manual_seed(2020)
P = torch.randn(U, d, device="cuda") * 0.1
L = rsx.lib()
for I in (100_000, 10_000, 1_000_000):
    Q = torch.randn(I, d, device="cuda") * 0.1
    G = torch.zeros_like(Q)
    for B in (65536, 1000000):
        u = torch.randperm(U, device="cuda", generator=gen)[:B].to(torch.int32)
        j = torch.randint(0, I, (B,), device="cuda", dtype=torch.int32, generator=gen)
        i = torch.randint(0, I, (B,), device="cuda", dtype=torch.int32, generator=gen)
        for mask, name in ((0, "full"), (8, "nt P load"), (16, "nt P store"), (24, "nt P load+store"), (3, "no atomics"), (3+24, "no atomics, nt")):
            L.rsx_debug_set_ablation(mask)
            t = timeit(lambda: rsx.bpr_step(P, Q, G, u, i, j, 0.05, 1.0 / B, users_unique=True))
            print(f"I={I} B={B} uniform {name}: {t*1e6:.1f}us  {B/t/1e6:.0f} M/s", flush=True)
L.rsx_debug_set_ablation(0)
