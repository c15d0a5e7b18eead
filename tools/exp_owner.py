"""timing experiment: blocked kernel at any batch (neg_block forced), sorted or not"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d = 1_000_000, 100_000, 128
dev = torch.device("cuda")
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev, popularity="zipf")
eng = BPREngine(P, Q, 0.05)
eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256, 16)
loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device=dev)
for B in (65536, 262144, 1_000_000):
    for nb in (0, 6, 8, 16):
        for sort in (False, True):
            if nb == 0 and sort: continue
            eng.neg_block = nb; eng._csr = None
            u, i, j = eng._triplet_buffers(B)
            kw = {}
            if nb or sort:
                ws = torch.empty(rsx.bpr_sample_workspace(B, I), dtype=torch.uint8, device=dev)
                eng._bind_csr(ip, ix)
                kw = dict(neg_block=nb, neg_key=12345, sort_pos=sort, ws=ws, user_sig=eng._sig, item_cdf=eng._cdf)
            try:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                rsx.bpr_sample(ip, ix, I, B, 7, 3, 0, u, i, j, **kw)
                torch.cuda.synchronize()
                a.record(); rsx.bpr_sample(ip, ix, I, B, 7, 3, 0, u, i, j, **kw); b.record(); torch.cuda.synchronize()
                ts_ = a.elapsed_time(b) * 1e3
            except Exception as e:
                print(B, nb, sort, "sampler:", str(e)[:100]); continue
            ts = []
            for rep in range(9):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rsx.bpr_step(P, Q, eng.G, u, i, j, 0.05, 1.0 / B, loss_acc=loss, users_unique=True, hot=eng.hot, neg_block=nb, neg_key=12345)
                b.record()
                rsx.apply_item_grad(Q, eng.G, 0.0, hot=eng.hot)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            print(f"{os.path.basename(rsx.LIB_PATH)} B={B} nb={nb} sorted={sort}: step kernel {sorted(ts)[4]:.1f} us (min {min(ts):.1f}); sampler {ts_:.0f} us")
