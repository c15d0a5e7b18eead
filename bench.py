#!/usr/bin/env python3
"""bench.py -- BPR triplet updates/s (+ full-catalog scores/s) on N MI355X of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic triplets, queued by the native
batch loop (include/rsx.h: rsx_bpr_trainer_run -- the reference's inner loop, models/MF.py:61-72):
  device sampling (rsx_bpr_sample, on a side stream, one step ahead) -> rsx_bpr_step
  -> [all-reduce of the item gradients over RCCL when N > 1] -> rsx_apply_item_grad.
Headline workload = BASELINE.json configs[2] (the d=128 shape the metric is quoted on): 1M users x
100K items per GPU, d=128, Zipf item popularity, 20 positives/user, one triplet per user per step
(the reference's epoch, data/generators.py:182-195), tables N(0, 0.1^2) resident in HBM before the
timed region.  Weak scaling: every rank owns its own 1M-user block (user rows sharded, items replicated).

Prints ONE JSON line (rank 0).
  roofline      dominant kernel of the headline leg.  `achieved` / `frac` are PHYSICAL: the HBM bytes one launch moves
                (`traffic`: rocprofv3 PMC counters, collected per leg by tools/refresh_profiles.sh into
                profiles/traffic.json -- `traffic_source` names the profile file and the commit it was taken at;
                the counters cannot be read inside this process) / the kernel's mean duration (HIP events on the
                launch stream around EVERY step kernel of the timed region) / the 8 TB/s peak.  A leg without a
                profiled traffic figure prints null and says why.  `compulsory_bytes` / `frac_compulsory`: what
                the batch cannot avoid moving once (P read + write, every touched Q row read, every touched G row
                read + written).  The contract's ALGORITHMIC figure (24*d bytes per triplet) is
                reported as `algorithmic_bytes_per_launch` / `algorithmic_GBs` / `algorithmic_rate_over_peak`:
                the blocked kernel sums item-side gradients on chip, so those bytes are not all moved and that
                ratio is NOT a fraction of anything (it exceeds 1 at the headline shape).
                `traffic_source.stale`: the kernel sources hash differently from the ones the profile was taken on.
                `roofline.configs`: every BASELINE config's {value, ms_per_step, kernel_ms, frac, frac_end_to_end} in compact form (the
                driver keeps `roofline`, `config` and `cpu_baseline` of this line and drops the rest).
  N > 1         RSX_EXCHANGE=allreduce (default: RCCL all-reduce issued by the library) | scatter_gather | direct (the library's own
                full mesh over HIP IPC / xGMI, include/rsx.h rsx_mesh_*: reduce-scatter + all-gather by direct peer reads);
                config.exchange_issued_by / config.exchange say which one ran.
  value         the MEDIAN of three back-to-back timed regions of --steps steps each (`timed_regions`: every region, min, max).
  hbm_utilisation_end_to_end   everything a step moves at the fabric side (step kernel + apply + the sampler beside them, PMC) over the
                whole step, as a fraction of the 8 TB/s peak: how busy the loop as a whole keeps HBM.
                roofline.achieved_over_copy_rate / hbm_utilisation_e2e_over_copy_rate: the same against 6.29 TB/s, what a float4 copy
                measures on this part (MI355X_MICROARCH.md) -- a second reference point; `peak` and `frac` stay on the 8 TB/s spec.
  legs          the other section-8d measurements, each with its own roofline: SURVEY's base batch
                B = 65 536, independent uniform negatives, uniform item popularity, a batch sweep, the
                configs[1] (d=64) shape, and the configs[3] one-rank slice (1.25M users x 1M items).
  scoring       full-catalog scores/s (MFMA fp32 roofline).
  cpu_baseline  the torch-CPU port of the reference path (oracle/torch_port.py; test infrastructure, used
                here only as the thing compared against) on this host: SGD and as-shipped Adam at the SAME
                batch as each GPU leg, the 1024 x I scoring tile and top-50.
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
HBM_COPY_GBS = 6290.0          # MI355X_MICROARCH.md: what a float4 copy measures on this part (79 % of the spec): NOT the roofline's `peak`,
                               # a second reference point only (`*_over_copy_rate`): how far from what HBM delivers to a pure stream
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
    ap.add_argument("--hot-replicas", type=int, default=0, help="private rows per popular item (a power of two); 0 = the engine's choice")
    ap.add_argument("--neg-block", type=int, default=8, help="item block of the stratified negatives (0 = independent uniform negatives)")
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("RSX_CHUNKS", "-1")),
                    help="> 1: every step as that many independent pipelines over item ranges (include/rsx.h: item chunks): the "
                         "all-reduce + apply of a range travel under the other ranges' kernels of this and the next step.  "
                         "-1 (default): 2 when N > 1 and the library issues the exchange, 0 (off) on one GPU, where there is "
                         "nothing to hide and the plain blocked step is faster")
    ap.add_argument("--no-legs", action="store_true", help="headline only (no section-8d legs)")
    ap.add_argument("--no-lightgcn", action="store_true", help="skip the BASELINE configs[4] leg (LightGCN propagation + step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def traffic_for(key):
    """PMC-measured HBM traffic of the leg's dominant kernel (profiles/traffic.json entry), if that leg was profiled"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:       # noqa: BLE001 -- no profile, no traffic figure
        return None


KERNEL_SOURCES = {"step": ("rsx_bpr.hip", "rsx_sample.hip", "rsx_train.hip", "rsx_common.h"), "spmm": ("rsx_graph.hip", "rsx_common.h")}


def sources_sha(kind):
    """hash of the kernel sources a profiled traffic figure depends on.  tools/install_profiles.py records it with every
    profiles/traffic.json entry; a line whose sources hash differently says `stale: true` (the .git directory does not travel
    to the GPU box: a commit cannot be compared there, file contents can)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[kind]:
        h.update(open(os.path.join(ROOT, "recsys_pytorch_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def traffic_source(t, kind, how):
    if not t:
        return None
    sha = t.get("sources_sha")
    return {"file": "profiles/" + str(t.get("profile")), "taken_at_commit": t.get("commit"), "how": how,
            # the kernels this figure was measured on vs. the ones in this tree (None: profile older than the hash record)
            "stale": (sha != sources_sha(kind)) if sha else None}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def roofline(kernel, kern_ms, n_timed, B, I, d, key, two_pass=False):
    """PHYSICAL roofline of one leg's step kernel: achieved = PMC-measured HBM bytes per launch / mean launch duration"""
    alg = (20 if two_pass else 24) * d * B          # per launch; the item pass writes no P row
    touched = min(2 * B, I)
    compulsory = 8 * d * B + 4 * d * touched + 8 * d * touched
    t = traffic_for(key)
    hbm = t.get("hbm_bytes_per_launch") if t else None
    per_s = lambda nbytes: nbytes / (kern_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "kernel_ms": kern_ms, "kernel_launches_timed": n_timed,
            "achieved": per_s(hbm) if hbm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": per_s(hbm) / HBM_PEAK_GBS if hbm else None, "traffic": hbm,
            "traffic_source": traffic_source(t, "step", "rocprofv3 --pmc passes over tools/step_prof.py at this leg's shape "
                                             "(tools/refresh_profiles.sh); read from profiles/traffic.json, NOT measured in this run"),
            **({} if hbm else {"frac_null_reason": f"no PMC profile of this leg in profiles/traffic.json (key {key})"}),
            "traffic_key": key,
            "compulsory_bytes": compulsory, "frac_compulsory": per_s(compulsory) / HBM_PEAK_GBS,
            "traffic_over_compulsory": hbm / compulsory if hbm else None,
            "algorithmic_bytes_per_launch": alg, "algorithmic_GBs": per_s(alg),
            "algorithmic_rate_over_peak": per_s(alg) / HBM_PEAK_GBS,
            # SURVEY section 8d's formula is a FRACTION only while the kernel really moves 24 d bytes per triplet
            "algorithmic_valid": per_s(alg) / HBM_PEAK_GBS <= 1.0,
            **({"algorithmic_invalid_reason": "the batch holds several triplets per item and the kernel sums item-side gradients on chip "
                                              "(positive runs in registers, negatives in a wave-private LDS tile): fewer than 24 d bytes per "
                                              "triplet reach HBM, so bytes/time exceeds the peak -- use frac (PMC bytes) or the "
                                              "independent-negatives leg (iid_*), where the formula holds"}
               if per_s(alg) / HBM_PEAK_GBS > 1.0 else {}),
            "note": "achieved / frac are physical (HBM bytes the launch moved, PMC).  algorithmic_* is SURVEY section 8d's "
                    "24 d bytes per triplet over the kernel time: where item sums stay on chip those bytes are not moved and "
                    "the ratio to the peak is not a fraction (it can exceed 1)"}


SHARDED = False     # a process group is up: N > 1, or RSX_FORCE_SHARDED=1 (the exchange path over a group of one rank)


def fence(world):
    torch.cuda.synchronize()
    if SHARDED:
        dist.barrier()
    torch.cuda.synchronize()


COMM = None         # rsx.Comm: the library's own RCCL communicator (N > 1 over the "nccl" backend, unless RSX_NATIVE_RCCL=0)


def mesh_selfcheck(eng, world, rank):
    """RSX_EXCHANGE=direct, N > 1: one exchange of the library's own mesh on known data, checked against the sum a torch.distributed
    all-reduce gives, before the leg is timed -- the mesh's ordering between GPUs (flags in uncached memory, system-scope fences,
    peer reads over xGMI) has only ever run between processes that share one GPU.  Leaves the tables as it found them.  Every rank
    learns every rank's verdict before anyone raises."""
    mesh, Qm, Gm = eng._mesh
    host = dist.get_backend() == "gloo"                  # (the one-GPU smoke runs: gloo reduces host tensors)

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if host:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)
    keep_Q = Qm.clone()
    rows = Qm.shape[0]
    gen = torch.Generator(device=Qm.device).manual_seed(777 + rank)
    err = None
    for rnd in range(3):                                 # three rounds: a stale line of round r would show in round r + 1
        Gm.copy_(torch.randn(Gm.shape, device=Gm.device, generator=gen) * (rnd + 1))
        want = Gm.clone()
        all_reduce(want)
        before = Qm.clone()
        torch.cuda.synchronize()
        dist.barrier()
        mesh.exchange_apply(0, rows, 0.5)
        try:
            mesh.check()
        except Exception as e:       # noqa: BLE001
            err = repr(e)
        ref = before - 0.5 * want
        bad = float((Qm - ref).abs().max()) if err is None else float("inf")
        if err is None and not (bad <= 1e-4 * float(want.abs().max()) and float(Gm.abs().max()) == 0.0):
            err = f"round {rnd}: |Q - (Q0 - lr * all_reduce(G))| max {bad:.3e}, |G| max {float(Gm.abs().max()):.3e} after the exchange"
        h = Qm.double().sum(1)                              # (collectives are unconditional: a rank with an error still takes part)
        cs = torch.stack([h, -h])
        all_reduce(cs, dist.ReduceOp.MAX)
        if err is None and not bool((cs[0] + cs[1] == 0.0).all()):
            err = f"round {rnd}: the ranks' item rows differ after the exchange"
        box = [None] * world
        dist.all_gather_object(box, err)
        if any(box):
            raise RuntimeError("mesh self-check failed before timing: " + "; ".join(f"rank {q}: {e}" for q, e in enumerate(box) if e))
    Qm.copy_(keep_Q)
    Gm.zero_()
    torch.cuda.synchronize()
    dist.barrier()


def step_leg(P, Q, indptr, indices, lr, B, want_nb, hot, hot_replicas, steps, warmup, world, rank, popularity, two_pass=None,
             chunks=0, regions=1):
    """`regions` back-to-back timed regions of `steps` native steps each (the MEDIAN region is the leg's figure); returns the
    leg record (rank 0 fills the throughput)"""
    from recsys_pytorch_amd import rsx
    from recsys_pytorch_amd.sharded import BPREngine
    U, d = P.shape
    I = Q.shape[0]
    dev = P.device
    B = min(B, U)
    # N > 1 exchange of the item gradients: all_reduce(G) (default) or RSX_EXCHANGE=scatter_gather
    # (reduce_scatter -> own item shard applied -> all_gather of the updated rows; sharded.py)
    eng = BPREngine(P, Q, lr, user_begin=rank * U, seed=2020, exchange=os.environ.get("RSX_EXCHANGE", "allreduce"),
                    force_sharded=SHARDED, comm=COMM)
    Q = eng.Q                                            # (scatter_gather may re-home the item table)
    if os.environ.get("RSX_CSC_SAMPLER", "0") == "1":    # A/B (opt-in): whole-pass batches sampled by the CSC walk instead of the bucket passes
        eng.use_csc = True
    if two_pass is not None:
        eng.overlap_exchange = bool(two_pass) and SHARDED and eng.exchange != "direct"      # (the mesh sums and applies in one go)
    # OPT-IN, never the default: RSX_STALE_EXCHANGE=1 lets a step's exchange travel under the NEXT step kernel (item
    # table one step stale: not the reference's batch-synchronous step; flagged in the JSON line)
    eng.stale_exchange = SHARDED and os.environ.get("RSX_STALE_EXCHANGE") == "1"
    nb = eng.set_neg_block(B, want_nb) if want_nb > 0 else 0
    if hot > 0:
        eng.set_hot_items(torch.bincount(indices.long(), minlength=I), hot, hot_replicas or None)
    if chunks > 1 and want_nb > 0 and not eng.stale_exchange and eng.exchange in ("allreduce", "direct"):
        # (B >= 2 I: blocked negatives inside the ranges; below: the ranges without blocks -- include/rsx.h "item chunks")
        eng.set_chunks(chunks)
        eng.overlap_exchange = False                     # the range pipeline replaces the two-pass step
        nb = eng.neg_block                               # (ranges use blocks of at least 3: sharded.py:pick_neg_block)
    if nb and os.environ.get("RSX_NEG_BLOCK_EXACT"):      # experiment: the block size itself, not pick_neg_block's choice
        nb = eng.neg_block = int(os.environ["RSX_NEG_BLOCK_EXACT"]); eng._csr = None; eng._relabel = None
    # the loss of every batch is accumulated on the device like MF.fit's epoch_loss (models/MF.py:70)
    loss = torch.zeros(rsx.RSX_LOSS_SLOTS, dtype=torch.float32, device=dev)
    tr = eng.native_trainer(indptr, indices, B, loss_acc=loss)
    gb = B * world
    if eng._mesh is not None and world > 1:
        mesh_selfcheck(eng, world, rank)                 # the library's own exchange against a torch.distributed all-reduce, BEFORE anything is timed
    tr.run(warmup, B, gb)
    if eng._mesh is not None:        # RSX_EXCHANGE=direct: a broken signal path shows in the warm-up already (bounded waits): stop here, loudly
        eng._mesh[0].check()
    loss.zero_()
    runs = []
    for _ in range(max(1, int(regions))):
        fence(world)
        t0 = time.perf_counter()
        # exactly `steps` steps inside the timed region, every step kernel timed (RSX_TIME_EVERY: development A/B of what the timing events cost)
        tr.run(steps, B, gb, time_every=int(os.environ.get("RSX_TIME_EVERY", "1")))
        fence(world)
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        if SHARDED:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        runs.append((float(el.item()),) + tuple(tr.kernel_ms()))
    order = sorted(range(len(runs)), key=lambda q: runs[q][0])
    elapsed, kern_ms, n_timed = runs[order[len(order) // 2]]            # the median region (and ITS kernel timings)
    region_ms = [r[0] / steps * 1e3 for r in runs]
    ran_chunks = getattr(tr, "chunks", 0)
    eng.adopt(tr)                                        # (a chunked run: checks it and copies the item rows back into Q)
    tr.close()
    mesh_exchanges = eng._mesh[0].info()[2] if eng._mesh is not None else None
    eng.close_mesh()                                     # (RSX_EXCHANGE=direct: collective -- no wait gave up, barrier, unmap the peers)
    assert torch.isfinite(P).all() and torch.isfinite(Q).all()
    mean_loss = float(loss.double().sum()) / (B * steps * len(runs))            # this rank's triplets
    assert np.isfinite(mean_loss) and 0.0 < mean_loss < 5.0, mean_loss
    replicas_equal = None
    if SHARDED:        # every rank applied the same reduced gradient: the item replicas must be identical, ROW BY ROW
        # (a keyed weighting of every row instead of one sum over the table: two rows swapped or a symmetric error in two
        #  ranks would cancel in a plain sum)
        gen = torch.Generator(device=dev).manual_seed(12345)
        w = torch.rand(Q.shape[1], device=dev, dtype=torch.float64, generator=gen) + 0.5
        h = (Q.double() * w).sum(1) * torch.arange(1, I + 1, device=dev, dtype=torch.float64)
        cs = torch.stack([h, -h])
        dist.all_reduce(cs, op=dist.ReduceOp.MAX)
        replicas_equal = bool((cs[0] + cs[1] == 0.0).all())        # max over ranks == min over ranks, for every item row
        assert replicas_equal, "item replicas diverged"
    # (the walk of the blocked kernel without its negative-side LDS tile when only the positives are ordered)
    kernel = "bpr_step_blocked_kernel" if nb else ("bpr_step_blocked_kernel<TILE=false>" if eng._sorts(B) else "bpr_step_kernel")
    key = f"U{U}_I{I}_d{d}_B{B}_{popularity}_nb{nb}" + (f"_c{ran_chunks}" if ran_chunks > 1 else "")
    return {"batch_per_gpu": B, "chunks": ran_chunks, "mesh_exchanges": mesh_exchanges,
            "sampler": ("CSC walk (whole-pass batch: one streaming pass over the transposed interactions, no buckets, no sort)"
                        if eng._csc is not None else "bucket passes (item-CDF buckets + LDS sort)" if eng._sorts(B, nb) else "plain"),
            "exchange_issued_by": (("library (own full mesh over HIP IPC / xGMI: rsx_mesh)" if eng.exchange == "direct" else
                                    "library (RCCL from librsx)" if COMM is not None else "torch.distributed callbacks")
                                   + (", range by range" if ran_chunks > 1 else "")) if SHARDED else None, "global_batch": gb, "value": gb * steps / elapsed, "unit": "triplets/s",
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "neg_block": nb, "mean_bpr_loss": mean_loss,
            "timed_regions": {"count": len(runs), "ms_per_step_each": region_ms, "min": min(region_ms), "max": max(region_ms),
                              "reported": "median"},
            "hot_replicas": eng.hot.replicas if eng.hot is not None else 0,
            "two_pass": bool(eng.overlap_exchange) and not eng.stale_exchange, "exchange": eng.exchange if SHARDED else None,
            "stale_exchange": bool(eng.stale_exchange),
            "item_replicas_identical": replicas_equal, "_Q": Q,
            "roofline": (rl := {**roofline(kernel, kern_ms, n_timed, B, I, d, key,
                                           two_pass=bool(eng.overlap_exchange) and not eng.stale_exchange),
                                **({"kernel_ms_is": f"sum of the {ran_chunks} item ranges' kernel durations per step, each timed on its "
                                                    "own stream (ranges that overlap on the chip count twice)"} if ran_chunks > 1 else {})}),
            # whole step (kernel + apply + gaps) against the same physical bytes of the step kernel
            "frac_end_to_end": (rl["traffic"] / (elapsed / steps) / 1e9 / HBM_PEAK_GBS) if rl["traffic"] else None,
            # ... and against EVERYTHING the step moves at the fabric side (PMC, same profile): the step kernel, the apply sweep and the
            # sampler of a later step that runs beside them -- how much of the HBM peak the loop as a whole keeps busy
            "hbm_utilisation_end_to_end": ((traffic_for(key) or {}).get("step_total_bytes") or 0) / (elapsed / steps) / 1e9 / HBM_PEAK_GBS or None,
            "step_total_bytes": (traffic_for(key) or {}).get("step_total_bytes"),
            "algorithmic_end_to_end_over_peak": gb / world * steps / elapsed * 24 * d / (HBM_PEAK_GBS * 1e9)}


def lightgcn_leg(U, I, d, indptr, indices, dev, layers=3, batch=65_536):
    """models/LightGCN.py:174-202 (propagation) and :83-87 (one training step: full-graph propagation, BPR gradient on
    the propagated tables, the same products on the gradient, dense Adam) at the BASELINE configs[4] shape.
    Roofline of the propagation product Y = A_hat X: algorithmic bytes nnz * (4 d + 8) + 2 N d 4."""
    import types
    import scipy.sparse as sp
    import recsys_pytorch_amd as pkg
    from recsys_pytorch_amd import rsx
    t0 = time.perf_counter()
    R = sp.csr_matrix((np.ones(indices.numel(), np.float32), indices.cpu().numpy(), indptr.cpu().numpy()), shape=(U, I))
    ds = types.SimpleNamespace(num_users=U, num_items=I, dataname="synthetic")
    m = pkg.LightGCN(ds, {"emb_dim": d, "num_layers": layers, "node_dropout": 0.0, "split": False, "num_folds": 1,
                          "reg": 0.0, "graph_dir": "graph"}, dev)
    g = m.getSparseGraph(R)
    build_s = time.perf_counter() - t0
    nnz, N = int(g.vals.numel()), U + I

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n
    t_spmm = timed(lambda: rsx.spmm(g, m._E0, m._ta, S_acc=m._out), 10)
    u = torch.randperm(U, device=dev)[:batch].int()
    i = torch.randint(0, I, (batch,), device=dev).int()
    j = torch.randint(0, I, (batch,), device=dev).int()
    t_step = timed(lambda: m.train_step(u, i, j), 5)
    assert torch.isfinite(m._E0).all()
    alg = nnz * (4 * d + 8) + 2 * N * d * 4
    compulsory = nnz * 8 + 2 * N * d * 4      # every embedding row once, the CSR once, the output once
    key = f"lightgcn_U{U}_I{I}_d{d}_L{layers}"
    t = traffic_for(key)
    hbm = t.get("hbm_bytes_per_launch") if t else None
    per_s = lambda nbytes: nbytes / t_spmm / 1e9
    return {"workload": f"BASELINE configs[4]: LightGCN {U} x {I}, d={d}, {layers} layers, nnz(A_hat)={nnz}",
            "propagation_ms_per_product": t_spmm * 1e3, "train_step_ms": t_step * 1e3, "batch": batch,
            "value": batch / t_step, "unit": "triplets/s",
            "graph_build_host_s": build_s,
            "roofline": {"bound": "hbm", "kernel": "spmm_csr_kernel", "kernel_ms": t_spmm * 1e3,
                         "achieved": per_s(hbm) if hbm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": per_s(hbm) / HBM_PEAK_GBS if hbm else None, "traffic": hbm,
                         "traffic_source": traffic_source(t, "spmm", "rocprofv3 --pmc passes over tools/spmm_prof.py; read from "
                                                          "profiles/traffic.json, NOT measured in this run"),
                         **({} if hbm else {"frac_null_reason": f"no PMC profile of this leg in profiles/traffic.json (key {key})"}),
                         "traffic_key": key,
                         "compulsory_bytes": compulsory, "frac_compulsory": per_s(compulsory) / HBM_PEAK_GBS,
                         # waste: what the product moves at the fabric side over what it must move (re-reads of neighbour rows)
                         "traffic_over_compulsory": hbm / compulsory if hbm else None,
                         "algorithmic_bytes_per_launch": alg, "algorithmic_GBs": per_s(alg),
                         "algorithmic_rate_over_peak": per_s(alg) / HBM_PEAK_GBS,
                         "algorithmic_valid": per_s(alg) / HBM_PEAK_GBS <= 1.0,
                         "gathered_bytes_served_on_chip": (alg - hbm) if hbm else None,
                         "note": "the algorithmic figure counts every gathered neighbour row (nnz rows of 4 d bytes); rows that are "
                                 "re-read are served by L2 / MALL (gathered_bytes_served_on_chip), the HBM side moves `traffic`"}}


def cpu_baseline(args, U, I, d, batches):
    """reference path (dense autograd + optimizer sweep) as ported in oracle/torch_port.py on this host,
    at the SAME batch as each GPU leg (SGD like the GPU path, and Adam as shipped, models/MF.py:30), plus the
    scoring tile P[users] @ Q.T (models/MF.py:109-112) and top-50 (python/func.py:4-17 via argpartition and
    func.h:12-31 via the C oracle).  Bounded to ~30 s of CPU work in all."""
    import oracle
    from oracle.torch_port import TorchMFPort
    cores = min(os.cpu_count() or 1, 64)   # dense fp32 sweeps stop scaling (and regress) past ~64 threads
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(2020)
    P0 = (torch.randn(U, d, generator=g) * 0.1).numpy()
    Q0 = (torch.randn(I, d, generator=g) * 0.1).numpy()
    rng = np.random.default_rng(1)
    legs = {}

    def timed(fn, budget, max_n):
        fn()
        ts, t_all = [], time.time()
        while len(ts) < max_n and (time.time() - t_all < budget or len(ts) < 2):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return float(np.median(ts)), len(ts)

    for opt, lr in (("sgd", args.lr), ("adam", 1e-3)):
        for B in batches:
            if opt == "adam" and B != batches[0]:
                continue
            m = TorchMFPort(P0, Q0, optimizer=opt, lr=lr)
            mk = lambda: (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
            med, n = timed(lambda: m.step(*mk()), 5.0, 20)
            legs[f"{opt}_B{B}"] = {"value": B / med, "unit": "triplets/s", "ms_per_step": med * 1e3, "steps": n}
            del m
    users = rng.integers(0, U, 1024)
    Pt, Qt = torch.from_numpy(P0), torch.from_numpy(Q0)
    score = lambda: (Pt[torch.from_numpy(users)] @ Qt.T).numpy()
    med, n = timed(score, 3.0, 10)
    legs["score_tile_1024xI"] = {"value": 1024 * I / med, "unit": "scores/s", "ms_per_tile": med * 1e3, "tiles": n}
    S = score().astype(np.float32)
    K = args.topk
    med, n = timed(lambda: np.argpartition(-S, K, axis=1)[:, :K], 3.0, 5)
    legs["top%d_numpy_argpartition" % K] = {"value": 1024 * I / med, "unit": "scores/s", "ms_per_tile": med * 1e3, "tiles": n}
    oracle.build(with_ref=False)
    med, n = timed(lambda: oracle.topk(S, K), 3.0, 5)
    legs["top%d_cxx_partial_sort_1_thread" % K] = {"value": 1024 * I / med, "unit": "scores/s", "ms_per_tile": med * 1e3, "tiles": n}
    head = legs[f"sgd_B{batches[0]}"]
    return {"value": head["value"], "unit": "triplets/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
            "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{head['steps']} SGD steps of B={batches[0]} (the headline batch) on U={U} I={I} d={d}: torch CPU port of "
                      f"models/MF.py:64-68 (dense grads + full optimizer sweep), median {head['ms_per_step']:.0f} ms/step; "
                      "other legs: same port at the other GPU batches, as-shipped Adam, scoring tile, top-k",
            "legs": legs}


def run_ranks(n, argv, extra_env, limit):
    """start N fresh ranks of this file (children: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous on a free port) and
    wait for them: (status, rank 0's JSON lines, what went wrong).  A failing rank ends the others -- exactly the processes started here."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = {**os.environ, "RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "RSX_LAUNCHED_BY": "bench.py", **extra_env}
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    lines = []

    def relay():                      # rank 0's stdout: JSON lines are collected, anything else goes to stderr
        for line in procs[0].stdout:
            (lines.append if line.startswith("{") else sys.stderr.write)(line)
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    t0, rc, why = time.time(), 0, None
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0 and rc == 0:
                    rc, why = c, f"rank {r} left with status {c}"
        if time.time() - t0 > limit and live:
            rc, why = 4, f"ranks {sorted(live)} still running after {limit:.0f} s"
        time.sleep(0.2)
    if rc != 0:                       # one rank failed: the others sit in a collective
        grace = time.time() + 10.0    # (their own watchdogs get a moment to print their error line)
        while time.time() < grace and any(p.poll() is None for p in procs):
            time.sleep(0.2)
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    t.join(timeout=10)
    return rc, lines, why


def launch_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the driver's plain command): start N fresh ranks of this
    file BEFORE this process has touched the GPU (it never does: it only relays), pass rank 0's single JSON line on, and leave with
    the first non-zero status of a rank.  Children, never a re-exec.  (The reference pins one device: main.py:24-27.)

    The exchange of the measured job is the default one (RCCL issued by the library).  So that the FIRST multi-GPU run also says what
    the library's own mesh over xGMI does (include/rsx.h: rsx_mesh_*; exact with two ranks on one GPU, never timed over real links),
    a second, short job of N fresh ranks then runs the headline alone with RSX_EXCHANGE=direct, in processes of its own -- whatever
    happens to it cannot touch the line of the first -- and its summary is merged into that line as `legs.exchange_direct_mesh`
    (+ `config.mesh_value / mesh_ms_per_step`).  RSX_BENCH_MESH_LEG=0 skips it."""
    limit = float(os.environ.get("RSX_LAUNCH_LIMIT_S", "3000"))
    rc, lines, why = run_ranks(n, sys.argv[1:], {}, limit)
    if rc != 0 or not lines:
        for line in lines:
            sys.stdout.write(line)
        if not lines:
            print(json.dumps({"metric": "bpr_triplet_updates_per_sec", "value": None, "unit": "triplets/s", "n_gpus": n,
                              "error": f"bench.py launcher: {why or 'rank 0 printed no JSON line'}"}), flush=True)
        sys.stdout.flush()
        return rc or 5
    out = lines[-1]
    if os.environ.get("RSX_BENCH_MESH_LEG", "1") != "0" and os.environ.get("RSX_EXCHANGE", "allreduce") != "direct":
        try:
            d = json.loads(out)
            argv = [a for a in sys.argv[1:]] + ["--no-legs", "--score-tiles", "0", "--no-cpu-baseline"]
            # (bounded on every level: a kernel's wait for a peer 10 s, a leg 120 s, the whole second job 300 s; the job checks the
            #  mesh against a torch.distributed all-reduce on known data BEFORE it times anything: bench.py mesh_selfcheck)
            rc2, lines2, why2 = run_ranks(n, argv, {"RSX_EXCHANGE": "direct", "RSX_WATCHDOG_S": os.environ.get("RSX_MESH_LEG_WATCHDOG_S", "120"),
                                                    "RSX_MESH_WAIT_S": os.environ.get("RSX_MESH_WAIT_S", "10")},
                                          float(os.environ.get("RSX_MESH_LEG_LIMIT_S", "300")))
            m = json.loads(lines2[-1]) if lines2 else {}
            if rc2 == 0 and m.get("value"):
                leg = {"value": m["value"], "unit": m["unit"], "ms_per_step": m["ms_per_step"], "kernel_ms": m["roofline"]["kernel_ms"],
                       "item_replicas_identical": m["config"].get("item_replicas_identical"), "item_chunks": m["config"].get("item_chunks"),
                       "mesh_exchanges": m["config"].get("mesh_exchanges"), "exchange_issued_by": m["config"].get("exchange_issued_by"),
                       "vs_default_exchange": m["value"] / d["value"] if d.get("value") else None}
            else:
                leg = {"value": None, "error": m.get("error") or why2 or f"status {rc2}"}
            d.setdefault("legs", {})["exchange_direct_mesh"] = leg
            d["config"]["mesh_value"] = leg.get("value")
            d["config"]["mesh_ms_per_step"] = leg.get("ms_per_step")
            d["config"]["mesh_note"] = ("the same headline in a second job of N fresh ranks with the library's own exchange over xGMI "
                                        "(RSX_EXCHANGE=direct): " + ("replicas identical" if leg.get("item_replicas_identical") else str(leg.get("error"))[:80]))
            out = json.dumps(d) + "\n"
        except Exception as e:       # noqa: BLE001 -- the second job must never cost the first one its line
            sys.stderr.write(f"bench.py launcher: mesh leg not merged: {e!r}\n")
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0


class Watchdog:
    """A hang must be impossible to miss: the N > 1 run sits under a deadline that is RE-ARMED at every leg boundary (a slow full
    run is not a stuck one) and names the leg it caught; on expiry rank 0 prints a JSON error line and every rank leaves with
    status 3 (os._exit: no re-exec, no GPU teardown from a process whose queues are stuck)"""

    def __init__(self, limit, rank, world, info):
        import threading
        self.limit, self.rank, self.world, self.info = limit, rank, world, info
        self.leg, self.deadline, self.done = "start", time.time() + limit, threading.Event()
        self.t = threading.Thread(target=self._bark, daemon=True)
        self.t.start()

    def arm(self, leg):
        self.leg, self.deadline = leg, time.time() + self.limit

    def stop(self):
        self.done.set()

    def _bark(self):
        while not self.done.wait(0.25 if self.limit > 0 else 0.0):
            if time.time() >= self.deadline:
                msg = {"metric": "bpr_triplet_updates_per_sec", "value": None, "unit": "triplets/s", "n_gpus": self.world,
                       "error": f"watchdog: rank {self.rank} did not finish leg '{self.leg}' within {self.limit:.0f} s (a collective "
                                "or a stream wait is stuck)", "config": self.info()}
                if self.rank == 0:
                    print(json.dumps(msg), flush=True)
                sys.stderr.write(msg["error"] + "\n")
                sys.stderr.flush()
                os._exit(3)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))        # (nothing above or in launch_ranks touches the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    global SHARDED
    SHARDED = world > 1 or os.environ.get("RSX_FORCE_SHARDED") == "1"
    watchdog = None
    if world > 1:       # started BEFORE the rendezvous and the communicator: a rank that dies inside the collective rsx_comm_create
        #                 (or never arrives) leaves the others blocked
        watchdog = Watchdog(float(os.environ.get("RSX_WATCHDOG_S", "900")), rank, world,
                            lambda: {"item_chunks": args.chunks, "exchange_issued_by": "library (RCCL from librsx)" if COMM is not None
                                     else "torch.distributed callbacks"})
        watchdog.arm("rendezvous + communicator")
    if SHARDED:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("RSX_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only to
        if backend == "nccl":                                   # smoke-test the N>1 code path on one GPU
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from recsys_pytorch_amd import rsx
    if os.environ.get("RSX_EXCHANGE_DELAY_US"):      # DEVELOPMENT library only (RSX_LIB=.../librsx_dev.so): a stand-in for the exchange's
        import ctypes                                 # time on the wire, for the one-rank schedule comparison (tools/exchange_model.sh)
        fn = rsx.lib().rsx_debug_set_exchange_delay
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
        assert fn(int(os.environ["RSX_EXCHANGE_DELAY_US"])) == 0
        if os.environ.get("RSX_EXCHANGE_TRAFFIC") == "1":     # ... and it moves the message through HBM while it holds the stream
            fn2 = rsx.lib().rsx_debug_set_exchange_traffic
            fn2.restype, fn2.argtypes = ctypes.c_int, [ctypes.c_int]
            assert fn2(1) == 0
    if os.environ.get("RSX_MESH_MODEL_WORLD"):       # DEVELOPMENT library only: a one-rank mesh moves what rank 0 of W ranks would move through
        import ctypes                                 # HBM, with a per-phase wire time (tools/exchange_model_schedules.sh)
        fn = rsx.lib().rsx_debug_set_mesh_model
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int]
        assert fn(int(os.environ["RSX_MESH_MODEL_WORLD"]), int(os.environ.get("RSX_MESH_MODEL_DELAY_US", "0"))) == 0
    if os.environ.get("RSX_SAMPLER_REPLAY", "0") != "0":  # DEVELOPMENT library only: 1 = the loop without a sampler beside it (3 batches
        import ctypes                                    # replayed), 2 = replayed steps with the sampler running into a shadow buffer
        fn = rsx.lib().rsx_debug_set_sampler_replay
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
        assert fn(int(os.environ["RSX_SAMPLER_REPLAY"])) == 0
        if os.environ.get("RSX_SAMPLE_ABLATION"):        # ... and parts of the sampler switched off (timing only: tools/sampler_parts.sh)
            fn = rsx.lib().rsx_debug_set_sample_ablation
            fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int]
            assert fn(int(os.environ["RSX_SAMPLE_ABLATION"])) == 0
    global COMM
    comm_note = None
    if SHARDED and os.environ.get("RSX_DIST_BACKEND", "nccl") == "nccl" and os.environ.get("RSX_NATIVE_RCCL", "1") == "1" \
            and os.environ.get("RSX_EXCHANGE", "allreduce") != "direct":
        # the exchange is then issued by librsx on the trainer's own stream: no interpreter in the timed region.  Creating the
        # communicator is collective: every rank tries, the ranks agree on the outcome, and if ANY rank failed (librccl not
        # loadable, init error) ALL fall back to the torch.distributed collectives handed in as callbacks -- the same schedule
        # (item ranges included: include/rsx.h exchange_range), the path the two-rank tests run
        try:
            COMM = rsx.Comm()
            ok = torch.ones(1, device=dev)
            COMM.all_reduce(ok)                         # one real collective through it, checked
            torch.cuda.synchronize()
            good = float(ok.item()) == float(world) and COMM.info() == (rank, world)
            err = None if good else f"all-reduce of ones over {world} ranks returned {float(ok.item())}"
        except Exception as e:       # noqa: BLE001
            good, err = False, repr(e)
        flag = torch.tensor([1.0 if good else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) != 1.0:
            if COMM is not None:
                try:
                    COMM.close()
                except Exception:    # noqa: BLE001
                    pass
            COMM = None
            comm_note = f"library RCCL communicator unavailable on some rank ({err or 'another rank failed'}): exchange through torch.distributed callbacks"
            if rank == 0:
                print("bench.py: " + comm_note, file=sys.stderr, flush=True)
    from recsys_pytorch_amd.data import synthetic_csr
    rsx.lib()
    if os.environ.get("RSX_STEP_WAVES"):        # experiment: resident wavefronts per SIMD of the blocked step kernel
        rsx.set_option("step_waves", int(os.environ["RSX_STEP_WAVES"]))
    if os.environ.get("RSX_APPLY_STREAM"):      # experiment: the ranges' applies on a stream of their own (include/rsx.h: "apply_stream")
        rsx.set_option("apply_stream", int(os.environ["RSX_APPLY_STREAM"]))
    if os.environ.get("RSX_TOUCHED_APPLY"):     # experiment: the row-marked apply of small batches never (0) / by the rule (1) / always (2)
        rsx.set_option("touched_apply", int(os.environ["RSX_TOUCHED_APPLY"]))
    if os.environ.get("RSX_MESH_BLOCKS"):
        rsx.set_option("mesh_blocks", int(os.environ["RSX_MESH_BLOCKS"]))
    if os.environ.get("RSX_SCORE_LANES"):
        rsx.set_option("score_lanes", int(os.environ["RSX_SCORE_LANES"]))

    def tables(U, I, d, degree, popularity):
        torch.manual_seed(2020 + rank)
        P = (torch.randn(U, d, device=dev) * 0.1).contiguous()          # this rank's user block
        torch.manual_seed(2020)
        Q = (torch.randn(I, d, device=dev) * 0.1).contiguous()          # replicated item table
        indptr, indices = synthetic_csr(U, I, degree, dev, seed=2020 + rank, popularity=popularity)
        return P, Q, indptr, indices

    U, I, d, B = args.users, args.items, args.dim, min(args.batch, args.users)
    # N > 1.  Two-pass step: the exchange runs under the user pass.  Measured over one-rank RCCL:
    # the split costs 165 us per step, so it pays as soon as the exposed exchange is longer than that -- 51 MB over
    # xGMI is at the very best 168 us with all seven links of N = 8 perfectly used, 0.33 / 0.7 ms at N = 4 / 2:
    # two passes at every N > 1 (RSX_TWO_PASS=0 for the one-pass schedule)
    two_pass = SHARDED and os.environ.get("RSX_TWO_PASS", "1") == "1"
    P, Q, indptr, indices = tables(U, I, d, args.degree, args.popularity)
    if args.chunks < 0:
        # N > 1: two item ranges (they beat three at every exchange length), with the library's RCCL or -- the
        # fallback above, and what tests/test_sharded_gloo.py runs with two ranks -- the per-range callbacks
        args.chunks = 2 if world > 1 else 0
    arm = watchdog.arm if watchdog is not None else (lambda leg: None)
    arm("headline")
    head = step_leg(P, Q, indptr, indices, args.lr, B, args.neg_block, args.hot, args.hot_replicas, args.steps, args.warmup,
                    world, rank, args.popularity, two_pass=two_pass, chunks=args.chunks, regions=3)
    Q = head.pop("_Q")

    # ---- the other section-8d legs: each its own timed region of the same native loop ------------------------
    def leg(*a, **k):
        r = step_leg(*a, **k)
        r.pop("_Q")
        return r

    legs = {}
    if not args.no_legs:
        short = max(10, args.steps // 2)
        if world == 1:
            base = 65_536      # SURVEY section 8d's base batch: below 2 triplets per item nothing is summed on chip
            if base < B:
                legs["base_batch_65536"] = leg(P, Q, indptr, indices, args.lr, base, args.neg_block, args.hot,
                                                    args.hot_replicas, 200, 10, 1, 0, args.popularity)
            if args.neg_block > 0:
                legs["independent_uniform_negatives"] = leg(P, Q, indptr, indices, args.lr, B, 0, args.hot, args.hot_replicas,
                                                                 short, 3, 1, 0, args.popularity)
            sweep = []
            for b in (256, 4_096, 16_384, 262_144):      # (256: the reference's default batch, config.py)
                if b < B:
                    r = leg(P, Q, indptr, indices, args.lr, b, args.neg_block, args.hot, args.hot_replicas, 100, 10, 1, 0,
                                 args.popularity)
                    sweep.append({**{k: r[k] for k in ("batch_per_gpu", "value", "ms_per_step", "neg_block")},
                                  "kernel_ms": r["roofline"]["kernel_ms"], "frac": r["roofline"]["frac"],
                                  "algorithmic_rate_over_peak": r["roofline"]["algorithmic_rate_over_peak"]})
            legs["batch_sweep"] = sweep
            other = "uniform" if args.popularity == "zipf" else "zipf"
            ip2, ix2 = synthetic_csr(U, I, args.degree, dev, seed=2020, popularity=other)
            legs[f"{other}_item_popularity"] = leg(P, Q, ip2, ix2, args.lr, B, args.neg_block, args.hot, args.hot_replicas,
                                                        short, 3, 1, 0, other)
            del ip2, ix2
            if d != 64:       # BASELINE configs[1]: the same shape at d=64
                P64, Q64 = (torch.randn(U, 64, device=dev) * 0.1), (torch.randn(I, 64, device=dev) * 0.1)
                legs["config1_d64"] = leg(P64, Q64, indptr, indices, args.lr, B, args.neg_block, args.hot, args.hot_replicas,
                                               short, 3, 1, 0, args.popularity)
                del P64, Q64
        # BASELINE configs[3] as each of its ranks sees it: 1.25M users x 1M items per GPU, 10 positives per
        # user, B = 1.25M per GPU (B < 2 I: the ordered batch without blocks); with N > 1 the 512 MB exchange per step travels
        # range by range under the other range's kernels (item ranges without blocks, like the headline's with them)
        if (args.users, args.items, args.dim) == (1_000_000, 100_000, 128):
            arm("config3_slice_1.25Mx1M")
            P4, Q4, ip4, ix4 = tables(1_250_000, 1_000_000, 128, 10, args.popularity)
            legs["config3_slice_1.25Mx1M"] = leg(P4, Q4, ip4, ix4, args.lr, 1_250_000, args.neg_block, args.hot,
                                                      args.hot_replicas, 10, 2, world, rank, args.popularity, two_pass=two_pass,
                                                      chunks=args.chunks)
            del P4, Q4, ip4, ix4

    # ---- BASELINE configs[4]: LightGCN on the same interaction graph, d=128, 3 layers -----
    if world == 1 and not args.no_legs and not args.no_lightgcn and (args.users, args.items, args.dim) == (1_000_000, 100_000, 128):
        legs["config4_lightgcn"] = lightgcn_leg(U, I, d, indptr, indices, dev)

    # ---- scoring leg (reported beside the headline; its own timed region) ---------------
    scoring = None
    # N > 1: every rank scores ITS users (user rows sharded, items replicated: no collective), all ranks at the same
    # time; the job's rate is all ranks' scores over the slowest rank's time
    if args.score_tiles > 0 and (rank == 0 or SHARDED):
        arm("scoring")
        tiles, K = args.score_tiles, args.topk
        users = torch.arange(1024 * tiles, device=dev, dtype=torch.int32) % U
        ws = torch.empty(rsx.lib().rsx_score_topk_workspace_d(users.numel(), I, d) // 4 + 64, dtype=torch.float32, device=dev)
        mask = (indptr, indices)
        rsx.score_topk(P, Q, users, K, mask=mask, ws=ws)        # warm-up pass (untimed)
        torch.cuda.synchronize()
        dts = []
        for _ in range(5):                                      # five whole calls, each its own timed region; the median is reported
            fence(world)
            t1 = time.perf_counter()
            top = rsx.score_topk(P, Q, users, K, mask=mask, ws=ws)
            torch.cuda.synchronize()
            el = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if SHARDED:
                dist.all_reduce(el, op=dist.ReduceOp.MAX)       # the slowest rank's time
            dts.append(float(el.item()))
        dt = sorted(dts)[2]
        n_scores = 1024 * tiles * I * world                     # whole job: every rank's tiles
        scoring = {"metric": "full_catalog_scores_per_sec", "value": n_scores / dt, "unit": "scores/s", "ms_per_1024_users": dt / tiles * 1e3,   # (per rank)
                   "sample": f"{tiles} tiles of 1024 users x {I} items" + (f" on each of {world} ranks (its own users)" if world > 1 else "")
                             + f", mask + top-{K} on device; median of 5 calls "
                             f"(min {min(dts)*1e3:.2f} ms, max {max(dts)*1e3:.2f} ms per call" + (", slowest rank" if world > 1 else "") + ")",
                   "n_gpus": world,
                   "roofline": {"bound": "mfma", "achieved": n_scores * 2 * d / dt / 1e12,
                                "peak": MFMA_F32_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                                "frac": n_scores * 2 * d / dt / 1e12 / (MFMA_F32_PEAK_TFLOPS * world)},
                   "topk_rows": int(top.shape[0])}
    arm("final barrier")
    if SHARDED:
        dist.barrier()
    if watchdog is not None:
        watchdog.stop()

    if rank == 0:
        nb = head["neg_block"]
        out = {
            "metric": "bpr_triplet_updates_per_sec", "value": head["value"], "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: BPRMF synthetic {U} users/GPU x {I} items, d={d}, "
                                   "on-device negative sampling, SGD",
                       "users_per_gpu": U, "items": I, "d": d, "batch_per_gpu": B, "global_batch": head["global_batch"],
                       "positives_per_user": args.degree, "item_popularity": args.popularity, "lr": args.lr,
                       "negatives": f"stratified by item block of {nb}, batch sorted by positive item" if nb else "independent uniform",
                       "sampler": "on device, two steps ahead on a lowest-priority side stream: " + head["sampler"],
                       "loop": "native (rsx_bpr_trainer_run): no interpreter between the kernels of the timed region",
                       "item_chunks": head["chunks"], "exchange_issued_by": head["exchange_issued_by"],
                       # what RCCL's own communicator says (rsx_comm_info), not what the environment asked for
                       "rccl_world": COMM.info()[1] if COMM is not None else None,
                       "exchange": head["exchange"], "mesh_exchanges": head["mesh_exchanges"],
                       "launched_by": os.environ.get("RSX_LAUNCHED_BY", "torch.distributed.run" if world > 1 else "python bench.py"),
                       "dist_backend": os.environ.get("RSX_DIST_BACKEND", "nccl") if SHARDED else None,
                       **({"exchange_note": comm_note} if comm_note else {}),
                       **({"negatives_with_item_ranges": "a position's negative is uniform over the real items of the item range its "
                           f"positive fell in (1/{head['chunks']} of the catalog under a seeded relabelling, redrawn between native "
                           "runs): NOT the same draw as the N = 1 line's blocks over the whole catalog"} if head["chunks"] > 1 else {}),
                       **({"DEBUG_exchange_delay_us": int(os.environ["RSX_EXCHANGE_DELAY_US"])} if os.environ.get("RSX_EXCHANGE_DELAY_US") else {}),
                       "mean_bpr_loss": head["mean_bpr_loss"],
                       **({"item_replicas_identical": head["item_replicas_identical"]} if SHARDED else {}),
                       "hot_items": args.hot, "hot_replicas": head["hot_replicas"],
                       "parallelism": (f"user-sharded x{world}, items replicated, "
                                       + ("1 all-reduce(G)/step" if head["exchange"] == "allreduce" else
                                          "direct full-mesh reduce-scatter(G) by peer reads + own item slice applied + all-gather(Q rows) by peer reads"
                                          if head["exchange"] == "direct" else
                                          "reduce-scatter(G) + own item shard applied + all-gather(Q rows) per step")
                                       + (", under the user pass of a two-pass step" if head["two_pass"] else "")
                                       + (", ONE STEP STALE (opt-in RSX_STALE_EXCHANGE: the exchange travels under the next step "
                                          "kernel; not the reference's synchronous step)" if head["stale_exchange"] else ""))
                                      if SHARDED else "single GPU"},
            "roofline": head["roofline"],
            "frac_end_to_end": head["frac_end_to_end"],
            "hbm_utilisation_end_to_end": head["hbm_utilisation_end_to_end"], "step_total_bytes": head["step_total_bytes"],
            "algorithmic_end_to_end_over_peak": head["algorithmic_end_to_end_over_peak"],
        }
        # every BASELINE config's figure where the driver keeps it (it stores `roofline`, `config` and `cpu_baseline` of this line
        # and drops the rest): value, ms per step, kernel ms, the PHYSICAL fraction of its roofline, the same end to end
        sig = lambda x: None if x is None else float(f"{x:.3g}")
        cfgs = {}
        for name, key in (("base_batch_65536", "base_batch_65536"), ("config1_d64", "config1_d64"), ("config3_slice", "config3_slice_1.25Mx1M")):
            if key in legs:
                g = legs[key]
                cfgs[name] = {"value": sig(g["value"]), "ms_per_step": sig(g["ms_per_step"]), "kernel_ms": sig(g["roofline"]["kernel_ms"]),
                              "frac": sig(g["roofline"]["frac"]), "frac_end_to_end": sig(g["frac_end_to_end"])}
        if "config4_lightgcn" in legs:
            g = legs["config4_lightgcn"]
            cfgs["config4_lightgcn"] = {"value": sig(g["value"]), "ms_per_step": sig(g["train_step_ms"]),
                                        "kernel_ms": sig(g["propagation_ms_per_product"]), "frac": sig(g["roofline"]["frac"]),
                                        "frac_end_to_end": None}
        if scoring is not None:
            cfgs["scoring"] = {"value": sig(scoring["value"]), "ms_per_step": sig(scoring["ms_per_1024_users"]), "kernel_ms": None,
                               "frac": sig(scoring["roofline"]["frac"]), "frac_end_to_end": sig(scoring["roofline"]["frac"])}
        if cfgs:
            out["roofline"] = {**out["roofline"], "configs": cfgs}
        # ... and the same figures once more as FLAT scalars (the driver's record keeps the scalars of `roofline` / `config` and
        # drops nested objects): every BASELINE config readable from BENCH_rNN.parsed alone
        flat = {"frac_e2e": sig(head["frac_end_to_end"]), "hbm_utilisation_e2e": sig(head["hbm_utilisation_end_to_end"]),
                # the same two figures against the float4-copy rate the guide measured (6.29 TB/s), beside -- never instead of -- the 8 TB/s peak
                "copy_rate_GBs": HBM_COPY_GBS,
                "achieved_over_copy_rate": sig(head["roofline"]["achieved"] / HBM_COPY_GBS) if head["roofline"].get("achieved") else None,
                "hbm_utilisation_e2e_over_copy_rate": sig(head["hbm_utilisation_end_to_end"] * HBM_PEAK_GBS / HBM_COPY_GBS)
                if head.get("hbm_utilisation_end_to_end") else None}
        for pre, name in (("base65536", "base_batch_65536"), ("d64", "config1_d64"), ("config3", "config3_slice")):
            if name in cfgs:
                flat.update({f"{pre}_value": cfgs[name]["value"], f"{pre}_ms_per_step": cfgs[name]["ms_per_step"],
                             f"{pre}_frac": cfgs[name]["frac"], f"{pre}_frac_e2e": cfgs[name]["frac_end_to_end"]})
        if "config4_lightgcn" in cfgs:
            flat.update({"lightgcn_ms_per_step": cfgs["config4_lightgcn"]["ms_per_step"], "lightgcn_value": cfgs["config4_lightgcn"]["value"],
                         "lightgcn_product_ms": cfgs["config4_lightgcn"]["kernel_ms"], "lightgcn_product_frac": cfgs["config4_lightgcn"]["frac"],
                         # WASTE, named: fabric-side bytes of one product over its compulsory bytes (every row, the CSR, the output once)
                         "lightgcn_traffic_over_compulsory": sig(legs["config4_lightgcn"]["roofline"].get("traffic_over_compulsory"))})
        if "independent_uniform_negatives" in legs:      # the one leg where SURVEY 8d's 24 d bytes per triplet ARE moved: its algorithmic figure is a fraction
            g = legs["independent_uniform_negatives"]
            flat.update({"iid_value": sig(g["value"]), "iid_ms_per_step": sig(g["ms_per_step"]), "iid_kernel_ms": sig(g["roofline"]["kernel_ms"]),
                         "iid_frac": sig(g["roofline"]["frac"]), "iid_algorithmic_frac": sig(g["roofline"]["algorithmic_rate_over_peak"]),
                         "iid_algorithmic_frac_e2e": sig(g["algorithmic_end_to_end_over_peak"])})
        if scoring is not None:
            flat.update({"scoring_value": cfgs["scoring"]["value"], "scoring_frac": cfgs["scoring"]["frac"],
                         "scoring_ms_per_1024_users": cfgs["scoring"]["ms_per_step"]})
        srcs = [x for x in [head["roofline"].get("traffic_source")] + [g.get("roofline", {}).get("traffic_source") for g in legs.values()
                                                                       if isinstance(g, dict)] if x]
        flat["traffic_stale"] = (any(x["stale"] is True for x in srcs) or (None if any(x["stale"] is None for x in srcs) else False)) if srcs else None
        flat["traffic_commit"] = (head["roofline"].get("traffic_source") or {}).get("taken_at_commit")
        out["roofline"] = {**out["roofline"], **flat}
        out["config"] = {**out["config"], **{"roofline_" + k: v for k, v in flat.items()}}
        out["timed_regions"] = head["timed_regions"]
        if legs:
            out["legs"] = legs
        if scoring is not None:
            out["scoring"] = scoring
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, U, I, d, [B] + ([65_536] if 65_536 < B else []))
        print(json.dumps(out), flush=True)
    if SHARDED:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
