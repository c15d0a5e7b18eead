"""tools/microbench.py -- quick kernel timings on one MI355X (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recsys_pytorch_amd import rsx


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def zipf_items(n, I, gen):
    w = 1.0 / torch.arange(1, I + 1, dtype=torch.float64, device="cuda")
    return torch.multinomial(w, n, replacement=True, generator=gen).to(torch.int32)


def main():
    print(json.dumps(rsx.device_info(0)))
    U, I = 1_000_000, 100_000
    gen = torch.Generator(device="cuda"); gen.manual_seed(2020)
    for d in (128, 64):
        P = torch.randn(U, d, device="cuda") * 0.1
        Q = torch.randn(I, d, device="cuda") * 0.1
        G = torch.zeros_like(Q)
        for B in (65536, 262144, 1048576 if U >= 1048576 else 1000000):
            B = min(B, U)
            u = torch.randperm(U, device="cuda", generator=gen)[:B].to(torch.int32)
            ju = torch.randint(0, I, (B,), device="cuda", dtype=torch.int32, generator=gen)
            for items in ("uniform", "zipf"):
                i = torch.randint(0, I, (B,), device="cuda", dtype=torch.int32, generator=gen) if items == "uniform" else zipf_items(B, I, gen)
                for layout in (0,):
                    t_step = timeit(lambda: rsx.bpr_step(P, Q, G, u, i, ju, 0.05, 1.0 / B, users_unique=True))
                    t_app = timeit(lambda: rsx.apply_item_grad(Q, G, 0.05))
                    def both():
                        rsx.bpr_step(P, Q, G, u, i, ju, 0.05, 1.0 / B, users_unique=True)
                        rsx.apply_item_grad(Q, G, 0.05)
                    t_both = timeit(both)
                    tps = B / t_both
                    print(f"d={d} B={B} items={items} layout={'vec4' if layout else 'strided'}: step {t_step*1e6:.1f}us apply {t_app*1e6:.1f}us both {t_both*1e6:.1f}us -> {tps/1e6:.1f} M triplets/s, "
                          f"alg {tps*24*d/1e12:.3f} TB/s = {tps*24*d/8e12*100:.1f}% of 8TB/s", flush=True)
            # general path (duplicate users allowed)
            ws = torch.zeros(rsx.bpr_step_workspace(U, B, d), dtype=torch.uint8, device="cuda")
            ud = torch.randint(0, U, (B,), device="cuda", dtype=torch.int32, generator=gen)
            t = timeit(lambda: rsx.bpr_step(P, Q, G, ud, ju, ju, 0.05, 1.0 / B, users_unique=False, ws=ws))
            print(f"d={d} B={B} general path (dup users): {t*1e6:.1f}us -> {B/t/1e6:.1f} M triplets/s", flush=True)
            G.zero_()
        # sampler
        deg = 20
        indptr = (torch.arange(U + 1, device="cuda", dtype=torch.int64) * deg)
        idx = torch.randint(0, I, (U, deg), device="cuda", dtype=torch.int32, generator=gen).sort(dim=1).values.reshape(-1).contiguous()
        B = 65536
        uo = torch.empty(B, dtype=torch.int32, device="cuda"); io = torch.empty_like(uo); jo = torch.empty_like(uo)
        t = timeit(lambda: rsx.bpr_sample(indptr, idx, I, B, 2020, 1, 0, uo, io, jo))
        print(f"d={d} sampler B={B}: {t*1e6:.1f}us", flush=True)
        # scoring
        users = torch.arange(1024, device="cuda", dtype=torch.int32)
        S = torch.empty(1024, I, device="cuda")
        t_s = timeit(lambda: rsx.score(P, Q, users, out=S), iters=10)
        t_m = timeit(lambda: rsx.score(P, Q, users, mask=(indptr, idx), out=S), iters=10)
        t_k = timeit(lambda: rsx.topk(S, 50), iters=10)
        ws = torch.empty(1024 * I, device="cuda")
        t_f = timeit(lambda: rsx.score_topk(P, Q, users, 50, mask=(indptr, idx), ws=ws), iters=10)
        n = 1024 * I
        print(f"d={d} score 1024x{I}: gemm {t_s*1e3:.3f}ms ({n*2*d/t_s/1e12:.1f} TF) +mask {t_m*1e3:.3f}ms topk50 {t_k*1e3:.3f}ms fused-call {t_f*1e3:.3f}ms -> {n/t_f/1e9:.1f} G scores/s "
              f"({n*2*d/t_f/157.3e12*100:.1f}% of MFMA fp32 peak)", flush=True)
        del P, Q, G, S, ws


if __name__ == "__main__":
    main()
