import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
U, I, d, B, c = 1_000_000, 100_000, 128, 1_000_000, 8
P = torch.zeros(8, d, device="cuda"); Q = torch.zeros(I, d, device="cuda")
ip, ix = synthetic_csr(U, I, 20, "cuda", seed=2020)
eng = BPREngine(P, Q, 0.05); eng.neg_block = c
u, i, j = eng.sample(ip, ix, B)
i = i.long()
p = torch.arange(B, device="cuda")
wave = (p * I // B) // c
b0 = (wave * c * B + I - 1) // I
b1 = ((wave + 1) * c * B + I - 1) // I
ln = (b1 - b0 + 1) // 2
group = wave * 2 + ((p - b0) >= ln).long()
pairs = torch.unique(group * I + i)
print("positions", B, "distinct items", torch.unique(i).numel(), "groups", torch.unique(group).numel(), "(group,item) runs", pairs.numel())
cnt = torch.bincount(i, minlength=I)
hot = torch.topk(torch.bincount(ix.long(), minlength=I), 256).indices
ishot = torch.zeros(I, dtype=torch.bool, device="cuda"); ishot[hot] = True
print("runs on hot items", int(ishot[(pairs % I)].sum()), "runs on cold", int((~ishot[(pairs % I)]).sum()))
