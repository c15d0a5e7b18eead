"""Stand-alone time of one whole-pass batch: the bucket passes (rsx_bpr_sample, RSX_SAMPLE_SORT_POS + item CDF) against the CSC walk
(rsx_bpr_sample_csc), alone on the GPU (beside the step kernel both stretch: that is what tools/ab.sh with RSX_CSC_SAMPLER=0/1 measures).

    python tools/sampler_csc_time.py [--users 1000000 --items 100000 --degree 20 --neg-block 2 --pop zipf --iters 30]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx                      # noqa: E402
from recsys_pytorch_amd.data import synthetic_csr       # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--degree", type=int, default=20)
    ap.add_argument("--neg-block", type=int, default=2)
    ap.add_argument("--pop", default="zipf")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    U, I, c = a.users, a.items, a.neg_block
    ip, ix = synthetic_csr(U, I, a.degree, "cuda", seed=2020, popularity=a.pop)
    u, i, j = (torch.empty(U, dtype=torch.int32, device="cuda") for _ in range(3))
    sig = rsx.build_signature(ip, ix, c) if c else None
    cdf = rsx.build_item_cdf(ip, ix, I)
    ws = torch.empty(rsx.bpr_sample_workspace(U, I), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    csc = rsx.Csc(ip, ix, I)
    e1.record()
    torch.cuda.synchronize()
    build_ms = e0.elapsed_time(e1)
    ws2 = torch.empty(csc.sample_ws_bytes, dtype=torch.uint8, device="cuda")
    step = [0]

    def bucket():
        step[0] += 1
        rsx.bpr_sample(ip, ix, I, U, 2020, step[0], 0, u, i, j, neg_block=c, neg_key=2 * step[0] + 1, sort_pos=True, ws=ws, user_sig=sig, item_cdf=cdf)

    def walk():
        step[0] += 1
        rsx.bpr_sample_csc(csc, ip, ix, I, 2020, step[0], u, i, j, neg_block=c, neg_key=2 * step[0] + 1, ws=ws2, user_sig=sig)

    out = {"users": U, "items": I, "degree": a.degree, "neg_block": c, "popularity": a.pop, "csc": csc.info(), "csc_build_ms": build_ms,
           "bucket_passes_us": timed(bucket, a.iters), "csc_walk_us": timed(walk, a.iters)}
    out["csc_stream_GBs"] = csc.info()["nnz"] * csc.info()["entry_bytes"] / out["csc_walk_us"] / 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
