import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import build as _b; os.environ["RSX_LIB"] = _b.build(dev=True)   # the -DRSX_ABLATE build (built here if run on the build host)
import torch
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from tools.microbench import timeit
U, I, d = 100_000, 100_000, 128
torch.manual_seed(0)
P = torch.randn(U, d, device="cuda") * 0.1
Q = torch.randn(I, d, device="cuda") * 0.1
mask = synthetic_csr(U, I, 20, "cuda", seed=2020)
L = rsx.lib()
for tiles in (64,):
    users = torch.arange(1024 * tiles, device="cuda", dtype=torch.int32)
    ws = torch.empty(L.rsx_score_topk_workspace(users.numel(), I) // 4 + 64, dtype=torch.float32, device="cuda")
    for m, name in ((0, "full"), (2, "hits detected, nothing stored"), (4, "no hits at all")):
        L.rsx_debug_set_score_ablation(m)
        t = timeit(lambda: rsx.score_topk(P, Q, users, 50, mask=mask, ws=ws), iters=5, warm=1)
        print(f"tiles={tiles} fused {name}: {t*1e3/tiles:.3f} ms per 1024 rows -> {1024*tiles*I/t/1e9:.1f} G scores/s", flush=True)
    L.rsx_debug_set_score_ablation(0)
    del ws
S = torch.empty(1024, I, device="cuda")
t = timeit(lambda: rsx.score(P, Q, users[:1024], out=S), iters=10); print(f"dense gemm: {t*1e3:.3f} ms")
