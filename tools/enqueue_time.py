"""tools/enqueue_time.py : host time to QUEUE a step of the native loop vs the GPU time of the step, per batch size"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d = 1_000_000, 100_000, 128
dev = torch.device("cuda")
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev, popularity="zipf")
for B in (1024, 4096, 16384, 65536, 1_000_000):
    eng = BPREngine(P, Q, 0.05)
    eng.set_neg_block(B, 8)
    eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256, 16)
    tr = eng.native_trainer(ip, ix, B)
    n = 2000 if B < 500_000 else 200
    tr.run(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.run(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host enqueue {(t1 - t0) / n * 1e6:.1f} us/step, step on the GPU {(t2 - t0) / n * 1e6:.1f} us/step")
    tr.close()
