"""tools/sampler_quality.py : does the ORDER the fast samplers give a batch cost ranking quality?

The headline number rests on a batch ordered by positive item whose negatives are stratified by item block (include/rsx.h:
rsx_bpr_sample, neg_block); the reference draws every negative independently and uniformly (data/generators.py:151-201).  Each user still
sees a uniformly distributed negative, but the draws of one batch are no longer independent of each other.  This script trains the same
model with the same number of steps on a planted-factor dataset of the headline's PROPORTIONS (one positive per user per step, batch =
users = 20 x items, a popular head) under

    iid      independent uniform negatives, the plain kernel's order         (the reference's sampler, on the device)
    blocked  ordered by positive item, negatives from blocks of c items       (the headline's layout; c by the headline's rule)
    ranges   the same inside two item ranges, relabelled every few epochs    (bench.py's N > 1 default)
    csc      the blocked layout, positives drawn by the walk over the transposed interactions (opt-in sampler of round 6)

for several seeds, and prints NDCG@10 / Recall@10 on held-out positives along the way: mean and standard deviation over the seeds
per arm.  `python tools/sampler_quality.py [--users U --items I --seeds S --epochs E]` on an MI355X; one JSON line per fit, a table last."""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

import recsys_pytorch_amd as pkg


def planted(U, I, k_true, n_pos, n_held, seed, dev="cuda"):
    """users x items affinities = planted factors + a popularity head + Gumbel noise; every user's n_pos best items are the
    positives, n_held of them (at random) held out"""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    A = torch.randn(U, k_true, device=dev, generator=g); B = torch.randn(I, k_true, device=dev, generator=g)
    pop = 0.8 * torch.log1p(torch.arange(I, device=dev).flip(0).float())
    train, held = [], []
    for lo in range(0, U, 8192):
        a = A[lo:lo + 8192]
        noise = -torch.log(-torch.log(torch.rand(a.shape[0], I, device=dev, generator=g).clamp_(1e-12, 1 - 1e-7)))
        top = torch.topk(a @ B.T + pop + 1.5 * noise, n_pos, dim=1).indices
        perm = torch.argsort(torch.rand(top.shape, device=dev, generator=g), dim=1)
        top = torch.gather(top, 1, perm)
        held.append(torch.sort(top[:, :n_held], dim=1).values.cpu().numpy()); train.append(torch.sort(top[:, n_held:], dim=1).values.cpu().numpy())
    mk = lambda rows, n: sp.csr_matrix((np.ones(U * n, np.float32), np.concatenate(rows).ravel(), np.arange(U + 1) * n), shape=(U, I))
    return mk(train, n_pos - n_held), mk(held, n_held)


class Recorder:
    """an evaluator that keeps every score dictionary it returned, and the seconds of TRAINING in front of it (wall clock since
    start() minus the time spent inside evaluate)"""
    def __init__(self, ev):
        self.ev, self.hist, self.t0, self.eval_s = ev, [], None, 0.0

    def start(self):
        torch.cuda.synchronize()
        self.t0, self.eval_s = time.perf_counter(), 0.0

    def evaluate(self, model):
        torch.cuda.synchronize()
        t = time.perf_counter()
        s = self.ev.evaluate(model)
        torch.cuda.synchronize()
        self.eval_s += time.perf_counter() - t
        self.hist.append({**{k: float(v) for k, v in s.items()}, "train_s": time.perf_counter() - self.t0 - self.eval_s})
        return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=100_000)
    ap.add_argument("--items", type=int, default=5_000)
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--rank", type=int, default=16, help="rank of the planted factors")
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--epochs", type=int, default=150)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--arms", default="iid,blocked,ranges,csc")
    ap.add_argument("--lr", type=float, default=0.1, help="step size per triplet (the model's lr is this times the batch: the batch mean of models/MF.py:105 undone)")
    ap.add_argument("--max-block", type=int, default=8, help="hparams['neg_block'] of the blocked arms: the largest item block the engine may pick")
    ap.add_argument("--force-block", type=int, default=0, help="the item block c itself instead of the engine's choice: hparams neg_block = neg_block_min = c")
    a = ap.parse_args()
    U, I = a.users, a.items
    tr, held = planted(U, I, a.rank, 20, 5, seed=42)
    ds = pkg.InteractionData(tr, held, held)
    ev = pkg.Evaluator(ds.valid_input, ds.valid_target, "holdout", [10])
    cfg = types.SimpleNamespace(batch_size=U, num_epochs=a.epochs, verbose=0, test_from=a.every, test_step=a.every)
    blk = dict(neg_block=a.force_block, neg_block_min=a.force_block) if a.force_block else dict(neg_block=a.max_block)
    arms = {"iid": dict(neg_block=0, chunks=0), "blocked": dict(chunks=0, **blk), "ranges": dict(chunks=2, **blk), "csc": dict(chunks=0, **blk)}
    res = {}
    for arm in a.arms.split(","):
        for seed in range(1, a.seeds + 1):
            torch.manual_seed(seed)
            m = pkg.MF(ds, dict(hidden_dim=a.dim, pointwise=False, loss_func="ce", lr=a.lr * U, seed=seed, **arms[arm]), "cuda")
            with torch.no_grad():
                m._P.mul_(0.1); m._Q.mul_(0.1)
            if arm == "csc":
                m._engine.use_csc = True
            rec = Recorder(ev)
            if seed == 1 and "untrained" not in res:
                res["untrained"] = {k: float(v) for k, v in ev.evaluate(m).items()}
            rec.start()
            m.fit(ds, cfg, evaluator=rec)
            eng = m._engine
            assert not (a.force_block and arm != "iid") or eng.neg_block == a.force_block, (eng.neg_block, a.force_block)
            line = {"arm": arm, "seed": seed, "neg_block": int(eng.neg_block), "chunks": int(eng.chunks), "csc": bool(getattr(eng, "use_csc", False)),
                    "hist": rec.hist}
            print(json.dumps(line), flush=True)
            res[(arm, seed)] = rec.hist
    print(f"# planted-factor dataset {U} users x {I} items, 15 train + 5 held-out positives per user, d = {a.dim}, batch = users, "
          f"{a.epochs} steps, lr {a.lr} x batch, {a.seeds} seeds" + (f", item block forced to {a.force_block}" if a.force_block else "") + "; untrained: " + ", ".join(f"{k} {v:.4f}" for k, v in sorted(res["untrained"].items())))
    metrics = sorted(res["untrained"])
    pick = [k for k in metrics if k.startswith("NDCG")] + [k for k in metrics if k.startswith("Recall")]
    for mt in pick[:2]:
        print(f"# {mt}: mean +- std over the seeds, after every {a.every} steps")
        for arm in a.arms.split(","):
            cols = []
            for t in range(len(res[(arm, 1)])):
                v = np.array([res[(arm, s)][t][mt] for s in range(1, a.seeds + 1)])
                cols.append(f"{v.mean():.4f} +- {v.std(ddof=1) if len(v) > 1 else 0.0:.4f}")
            print(f"#   {arm:8s} " + "   ".join(cols))
    print(f"# seconds of training (evaluation excluded; includes the fit's set-up: sampler tables, trainer) in front of each evaluation, mean over the seeds")
    for arm in a.arms.split(","):
        print(f"#   {arm:8s} " + "   ".join(f"{np.mean([res[(arm, s)][t]['train_s'] for s in range(1, a.seeds + 1)]):8.3f}" for t in range(len(res[(arm, 1)]))))


if __name__ == "__main__":
    main()
