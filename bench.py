#!/usr/bin/env python3
"""bench.py -- BPR triplet updates/s (+ full-catalog scores/s) on N MI355X of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic triplets:
  device sampling (rsx_bpr_sample, on a second stream, one step ahead) -> rsx_bpr_step
  -> [all-reduce of the item gradients over RCCL when N > 1] -> rsx_apply_item_grad.
Workload = BASELINE.json configs[2] (the d=128 shape the metric is quoted on):
1M users x 100K items per GPU, d=128, Zipf item popularity, 20 positives/user,
tables N(0, 0.1^2) resident in HBM before the timed region.  Weak scaling: every
rank owns its own 1M-user block (user rows sharded, items replicated).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel
(bpr_step_blocked_kernel, or bpr_step_kernel when the batch is < 2x the catalog): algorithmic bytes 24*d per triplet over the
kernel's average launch duration measured with HIP events on the launch stream.
`cpu_baseline` is the torch-CPU port of the reference path (oracle/torch_port.py)
timed on this host on a bounded sample; it is test infrastructure and is used here
only as the thing-compared-against.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: matrix FP32 peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--users", type=int, default=1_000_000, help="users PER GPU")
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1_000_000,
                    help="triplets per step PER GPU (default: one triplet per user per step, the reference's epoch, "
                         "data/generators.py:182-195)")
    ap.add_argument("--degree", type=int, default=20)
    ap.add_argument("--popularity", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--score-tiles", type=int, default=64,
                    help="1024-user tiles scored for scores/s (0 = skip); 64 tiles = 6.5e9 scores (SURVEY section 8d)")
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--hot", type=int, default=256, help="popular items whose gradient rows are replicated (0 = off)")
    ap.add_argument("--hot-replicas", type=int, default=16)
    ap.add_argument("--neg-block", type=int, default=8, help="item block of the stratified negatives (0 = independent uniform negatives)")
    ap.add_argument("--small-batch", type=int, default=65_536, help="extra leg at this batch size (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=65_536)
    return ap.parse_args()


def cpu_baseline(args):
    """reference path (dense autograd + optimizer sweep) as ported in oracle/torch_port.py,
    SGD like the GPU path, same U/I/d, bounded to ~10-30 s of CPU work."""
    from oracle.torch_port import TorchMFPort
    cores = min(os.cpu_count() or 1, 64)   # dense fp32 sweeps stop scaling (and regress) past ~64 threads
    torch.set_num_threads(cores)
    U, I, d, B = args.users, args.items, args.dim, args.cpu_batch
    g = torch.Generator().manual_seed(2020)
    P0 = (torch.randn(U, d, generator=g) * 0.1).numpy()
    Q0 = (torch.randn(I, d, generator=g) * 0.1).numpy()
    m = TorchMFPort(P0, Q0, optimizer="sgd", lr=args.lr)
    rng = np.random.default_rng(1)
    mk = lambda: (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
    for _ in range(2):
        m.step(*mk())
    times, t_all = [], time.time()
    while len(times) < 30 and time.time() - t_all < 15.0:    # ~15 s of CPU work, at least a few steps
        b = mk()
        t0 = time.time()
        m.step(*b)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": B / med, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} SGD steps of B={B} on U={U} I={I} d={d} (torch CPU port of models/MF.py:64-68, "
                      f"dense grads + full optimizer sweep), median {med*1e3:.0f} ms/step"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RSX_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only to
        if backend == "nccl":                                   # smoke-test the N>1 code path on one GPU
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.data import synthetic_csr
    from recsys_pytorch_amd.sharded import BPREngine
    rsx.lib()

    U, I, d, B = args.users, args.items, args.dim, min(args.batch, args.users)
    torch.manual_seed(2020 + rank)
    P = (torch.randn(U, d, device=dev) * 0.1).contiguous()          # this rank's user block
    torch.manual_seed(2020)
    Q = (torch.randn(I, d, device=dev) * 0.1).contiguous()          # replicated item table
    indptr, indices = synthetic_csr(U, I, args.degree, dev, seed=2020 + rank, popularity=args.popularity)
    eng = BPREngine(P, Q, args.lr, user_begin=rank * U, seed=2020)
    gb = B * world
    eng.use_item_cdf = os.environ.get("RSX_NO_CDF", "0") != "1"         # development knobs
    inline_sampler = os.environ.get("RSX_INLINE_SAMPLER", "0") == "1"
    neg_block = eng.set_neg_block(B, args.neg_block) if args.neg_block > 0 else 0
    if args.hot > 0:
        eng.set_hot_items(torch.bincount(indices.long(), minlength=I), args.hot, args.hot_replicas)

    # the sampler of step t+1 runs on a second HIP stream while step t computes (it reads only
    # the CSR); both are inside the timed region
    side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("RSX_SIDE_PRIORITY", "0")))
    if inline_sampler:
        side = torch.cuda.current_stream()
    bufs = [{"t": eng._triplet_buffers(B), "ready": None, "free": None, "key": 0} for _ in range(2)]
    state = {"cur": 0, "next_step": 0}

    def prefetch(slot):
        buf = bufs[slot]
        if buf["free"] is not None:
            side.wait_event(buf["free"])
        with torch.cuda.stream(side):
            buf["key"] = eng._launch_sample(indptr, indices, B, buf["t"], state["next_step"])
            buf["ready"] = torch.cuda.Event()
            buf["ready"].record(side)
        state["next_step"] += 1

    side.wait_stream(torch.cuda.current_stream())
    prefetch(0)
    # N > 1.  Two-pass step: the all-reduce runs under the user pass.  It pays
    # when the exchange takes longer than the user pass plus the sampler (~375 us): expected with
    # the 1 or 3 xGMI links of 2 or 4 GPUs, not with the 7 links of 8 (estimate; RSX_TWO_PASS overrides)
    two_pass = world > 1 and os.environ.get("RSX_TWO_PASS", "1" if world <= 4 else "0") == "1"
    # one all-reduce of the whole G by default: RCCL's bus bandwidth still rises with the message size
    # around 51 MB, which outweighs hiding the apply sweep behind later chunks (RSX_EXCHANGE_CHUNKS > 1)
    n_chunks = max(1, int(os.environ.get("RSX_EXCHANGE_CHUNKS", "1"))) if world > 1 else 1
    g_chunks = list(torch.chunk(eng.G, n_chunks, dim=0))
    q_chunks = list(torch.chunk(eng.Q, n_chunks, dim=0))

    loss_acc = torch.zeros(rsx.RSX_LOSS_SLOTS, dtype=torch.float32, device=dev)

    def one_step(ev=None):
        main = torch.cuda.current_stream()
        cur = state["cur"]
        buf = bufs[cur]
        main.wait_event(buf["ready"])
        if world == 1:
            prefetch(cur ^ 1)           # sampler of step t+1 runs beside step t's kernel
        u, i, j = buf["t"]
        use_hot = eng.hot is not None
        hot = eng.hot if use_hot else None
        # the loss of every batch is accumulated on the device like MF.fit's epoch_loss (models/MF.py:70)
        kw = dict(users_unique=True, hot=hot, neg_block=neg_block, neg_key=buf["key"], loss_acc=loss_acc)
        if ev is not None:
            ev[0].record()
        if world == 1:
            rsx.bpr_step(eng.P, eng.Q, eng.G, u, i, j, eng.lr, 1.0 / gb, **kw)
            if ev is not None:
                ev[1].record()
            rsx.apply_item_grad(eng.Q, eng.G, eng.lr, hot=hot)      # replicas folded inside the sweep
        else:
            # two passes over the same triplets (include/rsx.h RSX_ITEMS_ONLY / RSX_USERS_ONLY): the item
            # pass completes G, and the one exchange of the step -- all_reduce(G) over RCCL/xGMI -- then
            # travels under the user pass and the next step's sampler instead of behind the whole kernel
            rsx.bpr_step(eng.P, eng.Q, eng.G, u, i, j, eng.lr, 1.0 / gb, only="items" if two_pass else None, **kw)
            if ev is not None:
                ev[1].record()
            if use_hot:
                rsx.fold_hot_grad(eng.G, eng.hot)              # the all-reduce needs the folded G
            # (optionally in item-range chunks, so that the apply sweep of chunk k runs while the later
            #  chunks are still travelling)
            works = [dist.all_reduce(c, op=dist.ReduceOp.SUM, async_op=True) for c in g_chunks]
            after_items = torch.cuda.Event()
            after_items.record(main)
            side.wait_event(after_items)
            prefetch(cur ^ 1)                                  # sampler of step t+1 beside the exchange
            if two_pass:
                rsx.bpr_step(eng.P, eng.Q, eng.G, u, i, j, eng.lr, 1.0 / gb, only="users", **{**kw, "loss_acc": None})
            buf["free"] = torch.cuda.Event()
            buf["free"].record(main)
            for w, qc, gc in zip(works, q_chunks, g_chunks):
                w.wait()
                rsx.apply_item_grad(qc, gc, eng.lr)
        if world == 1:
            # (an event record between two kernels of this queue costs ~10 us of launch gap: with one
            #  GPU the triplet buffer is released after the apply sweep, not between kernel and sweep)
            buf["free"] = torch.cuda.Event()
            buf["free"].record(main)
        eng.step_count += 1
        state["cur"] = cur ^ 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    # HIP events around the step kernel of every 5th timed step (an event pair costs ~9 us of
    # launch gap on this queue, so bracketing every step would slow the thing being measured)
    ev_every = 1 if args.steps < 10 else 5
    events = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for s in range(0, args.steps, ev_every)}
    loss_acc.zero_()
    fence()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(events.get(s))
    fence()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events.values()]))   # bpr_step_kernel, HIP events
    assert torch.isfinite(P).all() and torch.isfinite(Q).all()
    mean_loss = float(loss_acc.double().sum()) / (B * args.steps)      # this rank's triplets
    assert np.isfinite(mean_loss) and 0.0 < mean_loss < 5.0, mean_loss
    replicas_equal = None
    if world > 1:      # every rank applied the same reduced gradient: the item replicas must be identical
        cs = torch.stack([Q.double().sum(), -Q.double().sum()])
        dist.all_reduce(cs, op=dist.ReduceOp.MAX)
        replicas_equal = bool(float(cs[0] + cs[1]) == 0.0)    # max(sum) == min(sum)
        assert replicas_equal, "item replicas diverged"

    # ---- small-batch leg: SURVEY section 8d's base batch (65 536 triplets/step), reported beside
    # the headline.  Below 2 triplets per item there is nothing to sum on chip: atomic path.
    small = None
    if world == 1 and args.small_batch > 0 and args.small_batch < B:
        eng_s = BPREngine(P, Q, args.lr, seed=2021)
        if args.hot > 0:
            eng_s.set_hot_items(torch.bincount(indices.long(), minlength=I), args.hot, args.hot_replicas)
        n_small = 200
        for _ in range(10):
            eng_s.sampled_step_overlapped(indptr, indices, args.small_batch, want_loss=False)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(n_small):
            eng_s.sampled_step_overlapped(indptr, indices, args.small_batch, want_loss=False)
        torch.cuda.synchronize()
        dts = time.perf_counter() - ts
        small = {"batch": args.small_batch, "value": args.small_batch * n_small / dts, "unit": "triplets/s",
                 "ms_per_step": dts / n_small * 1e3, "steps": n_small,
                 "path": "bpr_step_kernel (one atomic row update per item row touched, hot-item replicas)",
                 "frac_of_hbm_roofline": args.small_batch * n_small / dts * 24 * d / (HBM_PEAK_GBS * 1e9)}
        assert torch.isfinite(P).all() and torch.isfinite(Q).all()

    # ---- scoring leg (reported beside the headline; its own timed region) ---------------
    scoring = None
    if args.score_tiles > 0 and rank == 0:
        tiles, K = args.score_tiles, args.topk
        if os.environ.get("RSX_SCORE_LANES"):
            rsx.set_option("score_lanes", int(os.environ["RSX_SCORE_LANES"]))
        users = torch.arange(1024 * tiles, device=dev, dtype=torch.int32) % U
        ws = torch.empty(rsx.lib().rsx_score_topk_workspace(users.numel(), I) // 4 + 64, dtype=torch.float32, device=dev)
        mask = (indptr, indices)
        rsx.score_topk(P, Q, users, K, mask=mask, ws=ws)        # warm-up pass (untimed)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        top = rsx.score_topk(P, Q, users, K, mask=mask, ws=ws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        n_scores = 1024 * tiles * I
        scoring = {"metric": "full_catalog_scores_per_sec", "value": n_scores / dt, "unit": "scores/s",
                   "sample": f"{tiles} tiles of 1024 users x {I} items, mask + top-{K} on device",
                   "roofline": {"bound": "mfma", "achieved": n_scores * 2 * d / dt / 1e12,
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": n_scores * 2 * d / dt / 1e12 / MFMA_F32_PEAK_TFLOPS},
                   "topk_rows": int(top.shape[0])}
    if world > 1:
        dist.barrier()

    if rank == 0:
        value = gb * args.steps / elapsed
        alg_bytes = (20 if two_pass else 24) * d * B               # per launch; item pass: no P write
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-measured HBM bytes per launch, if profiled
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                key = f"U{U}_I{I}_d{d}_B{B}_{args.popularity}_nb{neg_block}"
                traffic = t.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "bpr_triplet_updates_per_sec", "value": value, "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: BPRMF synthetic {U} users/GPU x {I} items, d={d}, "
                                   "on-device negative sampling, SGD",
                       "users_per_gpu": U, "items": I, "d": d, "batch_per_gpu": B, "global_batch": gb,
                       "positives_per_user": args.degree, "item_popularity": args.popularity, "lr": args.lr,
                       "negatives": f"stratified by item block of {neg_block}, batch sorted by positive item" if neg_block else "independent uniform",
                       "sampler": "on device, overlapped on a second HIP stream",
                       "mean_bpr_loss": mean_loss,
                       **({"item_replicas_identical": replicas_equal} if world > 1 else {}),
                       "hot_items": args.hot, "hot_replicas": args.hot_replicas if args.hot > 0 else 0,
                       "parallelism": (f"user-sharded x{world}, items replicated, 1 all-reduce(G)/step"
                                       + (f" in {n_chunks} chunks pipelined with the apply sweep" if n_chunks > 1 else "")
                                       + (", under the user pass of a two-pass step" if two_pass else "")) if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "bpr_step_blocked_kernel" if neg_block else "bpr_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms},
        }
        if small is not None:
            out["small_batch"] = small
        if scoring is not None:
            out["scoring"] = scoring
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
