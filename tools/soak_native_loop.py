"""a long run of the native loop at the headline shape (and as item ranges): nothing drifts, hangs or leaves its range; the loss falls"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d, B = 1_000_000, 100_000, 128, 1_000_000
ip, ix = synthetic_csr(U, I, 20, "cuda", seed=2020, popularity="zipf")
for chunks, steps in ((0, 20_000), (2, 6_000)):
    torch.manual_seed(1)
    P, Q = torch.randn(U, d, device="cuda") * 0.1, torch.randn(I, d, device="cuda") * 0.1
    eng = BPREngine(P, Q, 0.05, seed=5)
    eng.set_neg_block(B, 8)
    eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256, 32)
    if chunks:
        eng.set_chunks(chunks)
    acc = torch.zeros(rsx.RSX_LOSS_SLOTS, device="cuda")
    tr = eng.native_trainer(ip, ix, B, loss_acc=acc)
    losses = []
    t0 = time.time()
    for part in range(10):
        acc.zero_()
        tr.run(steps // 10)
        torch.cuda.synchronize()
        losses.append(float(acc.sum()) / (B * (steps // 10)))
    dt = time.time() - t0
    tr.check()
    eng.adopt(tr)
    step, pos = tr.state()
    ok = bool(torch.isfinite(P).all()) and bool(torch.isfinite(eng.Q).all()) and step == steps and pos == steps * U and all(b <= a + 1e-6 for a, b in zip(losses, losses[1:]))
    print(f"chunks={chunks} steps={steps} {dt / steps * 1e6:.1f} us/step  loss per tenth {[round(x, 4) for x in losses]}  state {(step, pos)}  G left {float(eng.G.abs().max())}  ok={ok}", flush=True)
    tr.close()
