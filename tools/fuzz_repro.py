"""Replay ONE problem of tests/test_gpu_parity.py::test_step_kernels_on_random_shapes (found by tools/fuzz_campaign.sh) with its switches
turned one at a time, step by step against the oracle:   python tools/fuzz_repro.py <RSX_FUZZ_SEED> <trial> [multiplier]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
seed, trial = int(sys.argv[1]), int(sys.argv[2])
os.environ["RSX_FUZZ_SEED"] = str(seed)
os.environ["RSX_FUZZ_TRIALS"] = sys.argv[3] if len(sys.argv) > 3 else "4"
import oracle as oracle_mod                                     # noqa: E402  (a development tool: the checker, like the tests)
from conftest import fuzz                                       # noqa: E402
from recsys_pytorch_amd import rsx                              # noqa: E402
from test_gpu_parity import random_step_problems, run_step_problem   # noqa: E402

rng, trials = fuzz(777, 48)
pb = next(p for p in random_step_problems(rng, trials) if p["trial"] == trial)
print({k: v for k, v in pb.items() if k not in ("P0", "Q0", "steps", "hot")}, "hot", None if pb["hot"] is None else pb["hot"][1:])


def report(tag, pb):
    hot_items = []

    def watch(step, Q, G, hot, orc):
        if hot is not None and not hot_items:
            hot_items.append(set(hot.items.cpu().numpy().tolist()))
        upd = np.abs(orc.Q - pb["Q0"]).max()
        dQ = np.abs(Q.cpu().numpy() - orc.Q).max(axis=1)
        bad = np.flatnonzero(dQ > 1e-4 * upd)
        u, i, j, i_dev = pb["steps"][step]
        msg = []
        for b in bad[:6]:
            pos = np.flatnonzero(i_dev == b)
            msg.append(f"item {b} hot={bool(hot_items and b in hot_items[0])} pos at {pos[:6].tolist()} (n={len(pos)}) neg n={int((j == b).sum())} "
                       f"err/upd={dQ[b] / upd:.3f}")
        print(f"  [{tag}] step {step}: {len(bad)} bad item rows; " + "; ".join(msg), flush=True)
    try:
        P, Q, G, hot, orc, ctx = run_step_problem(rsx, oracle_mod, pb, watch)
        eP = np.abs(P.cpu().numpy() - orc.P).max() / np.abs(orc.P - pb["P0"]).max()
        print(f"  [{tag}] P error / update {eP:.2e}; G left {float(G.abs().max()):.1e}; ghot left {0.0 if hot is None else float(hot.ghot.abs().max()):.1e}")
    except AssertionError as e:
        print(f"  [{tag}] loss assertion: {e}")


report("as found", pb)
for name, change in (("32-bit offsets", dict(wide=False)), ("no hot map", dict(hot=None)), ("no sorted hint", dict(batch_sorted=False)),
                     ("no blocks", dict(c=0)), ("two calls", dict(two_calls=True)),
                     ("1 replica", dict(hot=None if pb["hot"] is None else (pb["hot"][0], pb["hot"][1], 1)))):
    report(name, dict(pb, **change))

# ---- the item gradient of every step, before the sweep: G + folded replicas against the fp64 sum --------------------------------------------
print("item gradients per step (G + replicas) against fp64:")
d, U, I, B, lr = pb["d"], pb["U"], pb["I"], pb["B"], pb["lr"]
P, Q = torch.from_numpy(pb["P0"]).cuda(), torch.from_numpy(pb["Q0"]).cuda()
G = torch.zeros_like(Q)
hot = rsx.HotItems(torch.from_numpy(pb["hot"][0]), pb["hot"][1], pb["hot"][2], d, "cuda") if pb["hot"] is not None else None
for step, (u, i, j, i_dev) in enumerate(pb["steps"]):
    live = i_dev >= 0
    P64, Q64 = P.double(), Q.double()
    ul, il, jl = (torch.from_numpy(x[live].astype(np.int64)).cuda() for x in (u, i, j))
    x = (P64[ul] * (Q64[il] - Q64[jl])).sum(1)
    g = -torch.sigmoid(-x) / max(int(live.sum()), 1)
    G64 = torch.zeros_like(Q64)
    G64.index_add_(0, il, g[:, None] * P64[ul])
    G64.index_add_(0, jl, -g[:, None] * P64[ul])
    ut, it, jt = (torch.from_numpy(x.astype(np.int32)).cuda() for x in (u, i_dev, j))
    rsx.bpr_step(P, Q, G, ut, it, jt, lr, 1.0 / max(int(live.sum()), 1), users_unique=pb["unique"], hot=hot, neg_block=pb["c"],
                 neg_key=pb["key"] if pb["c"] else 0, wide_offsets=pb["wide"], batch_sorted=pb["batch_sorted"])
    Gt = G.double().clone()
    if hot is not None:
        rep = hot.ghot.double().view(hot.n, hot.replicas, d).sum(1)
        Gt.index_add_(0, hot.items.long(), rep)
    err = (Gt - G64).abs().max(1).values
    bad = torch.nonzero(err > 1e-5 * G64.abs().max()).flatten().tolist()
    print(f"  step {step}: bad rows {bad[:8]}", [(b, float(err[b] / G64.abs().max()), float(G[b].abs().max()), float(G64[b].abs().max()),
                                               int(hot.slot[b]) if hot is not None else None) for b in bad[:4]])
    if hot is not None:
        for b in bad[:2]:
            s = int(hot.slot[b])
            if s >= 0:
                r = hot.ghot.view(hot.n, hot.replicas, d)[s]
                print("    replicas of item", b, "slot", s, "max per replica", r.abs().max(1).values.tolist(), "want", float(G64[b].abs().max()),
                      "G row", float(G[b].abs().max()))
    G_before = G.clone()
    rep_before = hot.ghot.clone().view(hot.n, hot.replicas, d) if hot is not None else None
    if hot is not None and step % 2 == 0:
        rsx.fold_hot_grad(G, hot)
        Gf = G.double()
        e2 = (Gf - G64).abs().max(1).values
        print("    after fold: bad rows", torch.nonzero(e2 > 1e-5 * G64.abs().max()).flatten().tolist()[:8], "ghot left", float(hot.ghot.abs().max()))
        rsx.apply_item_grad(Q, G, lr)
    else:
        rsx.apply_item_grad(Q, G, lr, hot=hot)
    nzr = torch.nonzero(G.abs().max(1).values > 0).flatten().tolist()
    print(f"    after the sweep of step {step}: rows of G left non-zero {nzr[:8]}",
          [(b, int((i_dev == b).sum()), int((j == b).sum()), int(hot.slot[b]) if hot is not None else None, float(G[b].abs().max()),
            torch.nonzero(G[b] != 0).flatten().tolist()[:6], int((G[b] != 0).sum())) for b in nzr[:3]])
    for b in nzr[:2]:
        print("      row", b, "G before the sweep", G_before[b][:4].tolist(), "replicas before", rep_before[int(hot.slot[b])][:, :4].tolist(),
              "left", G[b][:4].tolist())
