"""tools/score_bench.py : the bench's scoring leg (64 x 1024 users x 100K items, mask + top-50) and the dense 1024 x I
product, for the library RSX_LIB names (same-box A/B of scoring kernel variants)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
U, I, d, K, tiles = 1_000_000, int(os.environ.get("ITEMS", 100_000)), int(os.environ.get("DIM", 128)), 50, 64
dev = torch.device("cuda")
if os.environ.get("RSX_SCORE_LANES"):
    rsx.set_option("score_lanes", int(os.environ["RSX_SCORE_LANES"]))
torch.manual_seed(0)
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev)
users = torch.arange(1024 * tiles, device=dev, dtype=torch.int32)
ws = torch.empty(rsx.lib().rsx_score_topk_workspace(users.numel(), I) // 4 + 64, dtype=torch.float32, device=dev)
out = torch.empty(1024, I, device=dev)
top = rsx.score_topk(P, Q, users, K, mask=(ip, ix), ws=ws)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    top = rsx.score_topk(P, Q, users, K, mask=(ip, ix), ws=ws)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = min(ts)
td = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); rsx.score(P, Q, users[:1024], out=out); b.record(); torch.cuda.synchronize()
    td.append(a.elapsed_time(b) * 1e3)
n = 1024 * tiles * I
print(f"{os.path.basename(rsx.LIB_PATH)}: fused {dt*1e6/tiles:.1f} us per 1024 rows = {n/dt/1e9:.1f} G scores/s "
      f"({n*2*d/dt/1e12/157.3:.3f} of MFMA peak); checksum {int(top.long().sum())}; dense 1024xI {sorted(td)[2]:.1f} us")
