"""tools/sampler_ranges_probe.py : the chunked sampler ALONE (no step kernels, one stream) for C = 2, 3, 4, 6, 8: us per call, and how
evenly its buckets are filled (experiment)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d, B = 1_000_000, 100_000, 32, 1_000_000
dev = torch.device("cuda")
ip, ix = synthetic_csr(U, I, 20, dev, popularity="zipf")
P = torch.randn(U, d, device=dev) * 0.1
for C in [int(a) for a in sys.argv[1:]] or (2, 3, 4, 6, 8):
    Q = torch.randn(I, d, device=dev) * 0.1
    eng = BPREngine(P, Q, 0.05)
    eng.set_neg_block(B, 8)
    eng.set_chunks(C)
    nb = eng.neg_block
    r = eng._build_relabel(ip, ix)
    Ic = r["Ic"]
    u, i, j = (torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
    ws = torch.empty(rsx.bpr_sample_workspace(B, C * Ic), dtype=torch.uint8, device=dev)
    cpos = torch.zeros(C + 1, dtype=torch.int64, device=dev)
    call = lambda s: rsx.bpr_sample_chunked(ip, r["indices"], C * Ic, I, C, B, 7, s, 0, u, i, j, cpos, nb, 12345 + s, ws, r["cdf"], user_sig=r["sig"])
    for s in range(3): call(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for s in range(20): call(s)
    b.record(); torch.cuda.synchronize()
    print(f"C={C} c={nb} Ic={Ic}: {a.elapsed_time(b) * 1e3 / 20:.1f} us per chunked sample call; ranges {[int(x) for x in (cpos[1:] - cpos[:-1]).tolist()]}"
          f" skipped {(i < 0).sum().item()}")
