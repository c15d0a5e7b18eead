"""the first two sampled batches of a seeded engine in every layout: the same in every process?"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from recsys_pytorch_amd.sharded import BPREngine
from recsys_pytorch_amd.data import synthetic_csr
h = lambda t: hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:8]
out = []
for name, U, I, B, chunks, ordered, hot in (("blocked", 2400, 400, 1000, 0, False, 16), ("ordered", 2400, 782, 1000, 0, True, 0), ("plain", 2400, 782, 1000, 0, False, 8),
                                            ("ranges+blocks", 2400, 300, 1000, 3, False, 16), ("ranges", 2400, 782, 1000, 3, True, 0),
                                            ("headline-like", 200_000, 20_000, 200_000, 0, False, 256)):
    ip, ix = synthetic_csr(U, I, 7, "cuda", seed=3)
    torch.manual_seed(1)
    P, Q = torch.randn(U, 64, device="cuda") * 0.1, torch.randn(I, 64, device="cuda") * 0.1
    eng = BPREngine(P, Q, 0.05 * B, seed=11)
    eng.set_neg_block(B, 8)
    if ordered:
        eng.sorted_min_batch = 1
    if hot:
        eng.set_hot_items(torch.bincount(ix.long(), minlength=I), hot, 4)
    if chunks:
        eng.set_chunks(chunks)
    tr = eng.native_trainer(ip, ix, B)
    hs = []
    for _ in range(2):
        tr.run(1); torch.cuda.synchronize()
        hs += [h(x) for x in tr.last_batch()[:3]]
    out.append(name + ":" + "".join(x[:4] for x in hs))
    tr.close()
print("HASH", " ".join(out))
