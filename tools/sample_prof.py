"""tools/sample_prof.py B nb sorted reps : the sampler alone (for rocprofv3 --kernel-trace --stats)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
B, nb, srt, reps = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), int(sys.argv[4])
U, I, d = 1_000_000, 100_000, 128
dev = torch.device("cuda")
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev, popularity="zipf")
eng = BPREngine(P, Q, 0.05)
eng.neg_block = nb
u, i, j = eng._triplet_buffers(B)
ws = torch.empty(rsx.bpr_sample_workspace(B, I), dtype=torch.uint8, device=dev)
eng._bind_csr(ip, ix)
kw = dict(neg_block=nb, neg_key=12345, sort_pos=srt, ws=ws, user_sig=eng._sig, item_cdf=eng._cdf)
for r in range(reps):
    rsx.bpr_sample(ip, ix, I, B, 7, r, (r * B) % (U - B), u, i, j, **kw)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for r in range(reps):
    rsx.bpr_sample(ip, ix, I, B, 7, r, (r * B) % (U - B), u, i, j, **kw)
b.record(); torch.cuda.synchronize()
print(f"B={B} nb={nb} sorted={srt}: {a.elapsed_time(b) * 1e3 / reps:.1f} us per sample call (back to back)")
