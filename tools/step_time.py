"""tools/step_time.py [B] [neg_block] : the step kernel of the bench workload stand-alone (HIP events, median
of 9) and the native loop per step (sampler on the side stream || step -> apply), for the library RSX_LIB names"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nbw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
U, I, d = int(os.environ.get("USERS", 1_000_000)), int(os.environ.get("ITEMS", 100_000)), int(os.environ.get("DIM", 128))
dev = torch.device("cuda")
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, int(os.environ.get("DEG", 20)), dev, popularity=os.environ.get("POP", "zipf"))
eng = BPREngine(P, Q, 0.05)
if os.environ.get("SORTED_MIN"):
    eng.sorted_min_batch = int(os.environ["SORTED_MIN"])
nb = eng.set_neg_block(B, nbw) if nbw else 0
eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256, 16)
u, i, j = eng.sample(ip, ix, B)
kw = dict(users_unique=True, hot=eng.hot, neg_block=nb, neg_key=eng.last_neg_key, batch_sorted=eng._sorts(B) and not nb)
loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device=dev)
for with_loss in (True, False):
    ts = []
    for rep in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rsx.bpr_step(P, Q, eng.G, u, i, j, 0.05, 1.0 / B, loss_acc=loss if with_loss else None, **kw)
        b.record()
        rsx.apply_item_grad(Q, eng.G, 0.0, hot=eng.hot)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"{os.path.basename(rsx.LIB_PATH)} B={B} nb={nb} step kernel alone ({'with' if with_loss else 'no'} loss): {sorted(ts)[4]:.1f} us  (min {min(ts):.1f})")
tr = eng.native_trainer(ip, ix, B, loss_acc=loss)
n = 200 if B >= 500_000 else 1000
tr.run(20)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.run(n, time_every=5)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"  native loop: {dt*1e6:.1f} us/step = {B/dt/1e9:.3f} G triplets/s; step kernel in the loop {tr.kernel_ms()[0]*1e3:.1f} us")
