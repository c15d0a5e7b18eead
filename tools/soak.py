import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d, B = 1_000_000, 100_000, 128, 1_000_000
dev = torch.device("cuda")
torch.manual_seed(0)
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev, popularity="zipf")
eng = BPREngine(P, Q, 20.0)          # large lr: the 1/B-scaled gradients then move the tables visibly
eng.set_neg_block(B, 8); eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256)
loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device=dev)
tr = eng.native_trainer(ip, ix, B, loss_acc=loss)
t0 = time.perf_counter()
for chunk in range(10):
    loss.zero_(); tr.run(300); torch.cuda.synchronize()
    print(f"steps {300*(chunk+1)}: mean BPR loss of the chunk {float(loss.double().sum())/(300*B):.4f}; |P|max {float(P.abs().max()):.3f} |Q|max {float(Q.abs().max()):.3f} finite {bool(torch.isfinite(P).all() and torch.isfinite(Q).all())}")
print(f"{3000/(time.perf_counter()-t0):.0f} steps/s")
