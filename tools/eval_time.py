"""tools/eval_time.py : what one Evaluator.evaluate costs at the headline shape (1M users x 100K items, d = 128; evaluation/evaluator.py:26-39 is
the call it mirrors) and where the time goes (cProfile).  python tools/eval_time.py [K]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import recsys_pytorch_amd as pkg
from recsys_pytorch_amd.data import synthetic_csr
U, I, d = 1_000_000, 100_000, 128
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ip, ix = synthetic_csr(U, I, 20, "cuda", seed=2020)
R = sp.csr_matrix((np.ones(U * 20, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
rng = np.random.default_rng(1)
T = sp.csr_matrix((np.ones(U * 5, np.float32), rng.integers(0, I, U * 5), np.arange(U + 1) * 5), shape=(U, I))
ds = pkg.InteractionData(R, T, T)
m = pkg.MF(ds, {"hidden_dim": d, "pointwise": False, "loss_func": "ce", "lr": 0.05, "optimizer": "sgd"}, "cuda")
ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "holdout", [10, K])
ev.evaluate(m); torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter(); s = ev.evaluate(m); torch.cuda.synchronize(); print(f"evaluate: {(time.perf_counter() - t) * 1e3:.1f} ms  ({U * I / (time.perf_counter() - t):.3g} scores/s end to end)")
pr = cProfile.Profile(); pr.enable(); ev.evaluate(m); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
# the scoring calls alone, chunk by chunk (is the time the kernel's, and does it depend on which users?)
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import csr_to_device
mask = csr_to_device(ds.valid_input, "cuda")
ws = torch.empty(rsx.lib().rsx_score_topk_workspace_d(65536, I, d) // 4 + 64, dtype=torch.float32, device="cuda")
for name, (P_, Q_) in (("model tables N(0,1)", (m._P, m._Q)), ("0.1 x", (m._P * 0.1, m._Q * 0.1))):
    ts = []
    for c in (0, 0, 1, 7, 15):
        users = torch.arange(c * 65536, min((c + 1) * 65536, U), device="cuda", dtype=torch.int32)
        torch.cuda.synchronize(); t = time.perf_counter(); rsx.score_topk(P_, Q_, users, K, mask=mask, ws=ws); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(name, "ms per 65536-user call, chunks 0 0 1 7 15:", [round(x, 1) for x in ts])
users = torch.arange(65536, device="cuda", dtype=torch.int32)
for kk in (10, 50):
    torch.cuda.synchronize(); t = time.perf_counter(); rsx.score_topk(m._P, m._Q, users, kk, mask=mask, ws=ws); torch.cuda.synchronize(); print("K", kk, round((time.perf_counter() - t) * 1e3, 1), "ms")
torch.cuda.synchronize(); t = time.perf_counter(); rsx.score_topk(m._P, m._Q, users, K, mask=None, ws=ws); torch.cuda.synchronize(); print("no mask", round((time.perf_counter() - t) * 1e3, 1), "ms")
# ... and the same calls as predict_topk makes them
real = rsx.score_topk
def timed(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = real(*a, **k); torch.cuda.synchronize(); timed.ts.append(round((time.perf_counter() - t) * 1e3, 1)); return r
timed.ts = []
m._k.score_topk = timed
t = time.perf_counter(); m.predict_topk(np.arange(U), ds.valid_input, K); print("predict_topk", round((time.perf_counter() - t) * 1e3, 1), "ms; per call", timed.ts)
# where do the slow calls come from?  the same 16 calls: back to back; with the D2H copy of the result in between; the copy into pinned memory
import gc
def loop(kind):
    ts = []
    pinned = torch.empty((65536, K), dtype=torch.int32).pin_memory() if kind == "pinned" else None
    for c in range(16):
        users = torch.arange(c * 65536, min((c + 1) * 65536, U), device="cuda", dtype=torch.int32)
        torch.cuda.synchronize(); t = time.perf_counter(); r = real(m._P, m._Q, users, K, mask=mask, ws=ws); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t) * 1e3, 1))
        if kind == "cpu":
            keep.append(r.cpu().numpy())
        elif kind == "pinned":
            pinned[:r.shape[0]].copy_(r); torch.cuda.synchronize(); keep.append(pinned[:r.shape[0]].numpy().copy())
    return ts
for kind in ("back to back", "cpu", "pinned", "cpu"):
    keep = []
    print(kind, loop(kind))
gc.disable(); keep = []; print("cpu, gc off", loop("cpu")); gc.enable()
# bisect: predict_topk's own loop, piece by piece
def loop2(users_from_numpy, np_concat):
    ts, out = [], []
    eu = np.arange(U)
    for s in range(0, U, 65536):
        users = m._idx(eu[s:s + 65536]) if users_from_numpy else torch.arange(s, min(s + 65536, U), device="cuda", dtype=torch.int32)
        torch.cuda.synchronize(); t = time.perf_counter(); r = real(m._P, m._Q, users, K, mask=mask, ws=ws); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t) * 1e3, 1))
        out.append(r.cpu().numpy())
    if np_concat:
        np.concatenate(out)
    return ts
print("users from numpy", loop2(True, False))
print("users on device ", loop2(False, False))
ws2 = torch.empty(rsx.lib().rsx_score_topk_workspace_d(65536, I, d) // 4 + 64, dtype=torch.float32, device="cuda")
ws_keep = ws; ws = ws2
print("fresh ws, numpy  ", loop2(True, False))
