"""tools/exp_config3_apply.py : the configs[3] one-rank slice (1.25M users x 1M items, d = 128, B = 1.25M): what a touched-row apply could
save over the dense sweep -- the share of item rows a step touches, the dense sweep's time alone, and the bytes
both forms must move."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d, B = 1_250_000, 1_000_000, 128, 1_250_000
dev = "cuda"
ip, ix = synthetic_csr(U, I, 10, dev, seed=2020)
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
eng = BPREngine(P, Q, 0.05, seed=2020)
eng.set_neg_block(B, 8)
eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256, None)
for t in range(3):
    u, i, j = eng.sample(ip, ix, B)
    eng.step_count += 1
    live = i >= 0
    ti, tj = torch.unique(i[live]).numel(), torch.unique(j[live]).numel()
    tu = torch.unique(torch.cat([i[live], j[live]])).numel()
    print(f"step {t}: live {int(live.sum())}  distinct positives {ti} ({ti / I * 100:.1f} % of the rows)  distinct negatives {tj} ({tj / I * 100:.1f} %)  "
          f"rows touched {tu} ({tu / I * 100:.1f} %)")
# the dense sweep on a gradient buffer with exactly those rows non-zero
G = torch.zeros_like(Q)
rows = torch.unique(torch.cat([i[live], j[live]])).long()


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        G[rows] = 1e-3
        torch.cuda.synchronize()
        t = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] * 1e6


t_dense = timed(lambda: rsx.apply_item_grad(Q, G, 0.05))
touched = rows.numel()
dense_bytes = I * d * 4 + 3 * touched * d * 4             # G read everywhere; Q read + Q written + G cleared where a gradient sits
sparse_bytes = 4 * touched * d * 4 + touched * 4 + I      # the same four row accesses on the touched rows + a row list / flag bytes
print(f"dense sweep: {t_dense:.0f} us for {dense_bytes / 1e9:.2f} GB = {dense_bytes / t_dense / 1e6:.2f} TB/s;  a touched-row apply would move "
      f"{sparse_bytes / 1e9:.2f} GB ({(1 - sparse_bytes / dense_bytes) * 100:.1f} % less) -- at the same rate {sparse_bytes / (dense_bytes / t_dense):.0f} us")
