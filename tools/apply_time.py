"""tools/apply_time.py : stand-alone time of rsx_apply_item_grad on the bench shape, dense and sparse gradient"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
I, d = 100_000, 128
Q = torch.randn(I, d, device="cuda")
for frac in (1.0, 0.5):
    ts = []
    for rep in range(10):
        G = torch.randn(I, d, device="cuda")
        if frac < 1.0:
            G[torch.rand(I, device="cuda") > frac] = 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rsx.apply_item_grad(Q, G, 1e-9); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"rows with a gradient {frac:.0%}: {sorted(ts)[len(ts) // 2]:.1f} us")
