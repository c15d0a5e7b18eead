"""Stress of rsx_mesh setup / exchange / teardown with W processes on ONE GPU (round 6: the driver's round-5 box failed
`test_mesh_exchange_on_small_and_odd_tables[4]` in rsx_mesh_local on rank 2 of 4).

    python tools/mesh_stress.py --world 4 --loops 20 [--no-post-barrier] [--own-memory] [--out gpurun_out/x.json]

Every loop builds FOUR meshes in a row over fresh small tables (the failing test's shapes: the first three come out of the SAME
pooled 2 MB segment of torch's allocator, so the same allocation is exported again and again), runs three exchanges through each and
checks Q against the torch sum, the replicas against each other, G == 0.
  --no-post-barrier   tear down the way round 5 did (barrier -> destroy, nothing after): the suspected race
  --own-memory        tables in memory of their own (one hipMalloc per table through torch's allocator with the caching pool
                      switched off for the allocation: PYTORCH_NO_HIP_MEMORY_CACHING is process wide, so the tool just pads the
                      tables to > 20 MB, which torch serves by an allocation of their own)
Prints one JSON line: loops run, failures (rank, loop, shape, message), export retries that were needed.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = ((1, 32), (5, 64), (257, 128), (1000, 256))


def worker(rank, world, port, loops, post_barrier, own_memory, out, empty_cache=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recsys_pytorch_amd import rsx
    dev = torch.device("cuda", 0)
    fails, retries, meshes = [], 0, 0
    phases = {}
    t0 = time.time()
    for loop in range(loops):
        for rows, d in SHAPES:
            gen = torch.Generator().manual_seed(1000 * rows + d + 7 * loop)
            Q0 = torch.randn(rows, d, generator=gen)
            Gs = [torch.randn(rows, d, generator=gen) for _ in range(world)]
            if own_memory == 1:                  # > 20 MB: an allocation of its own, not a piece of a pooled segment
                pad = (24 << 20) // (4 * d) + 1
                Qb = torch.zeros(pad, d, device=dev); Gb = torch.zeros(pad, d, device=dev)
                Q, G = Qb[:rows], Gb[:rows]
                Q.copy_(Q0); G.copy_(Gs[rank])
            elif own_memory == 2:                # memory from rsx_mesh_alloc (exported at allocation, handle cached)
                Q, G = rsx.mesh_tensor(rows, d), rsx.mesh_tensor(rows, d)
                Q.copy_(Q0); G.copy_(Gs[rank])
            else:
                Q, G = Q0.clone().to(dev), Gs[rank].clone().to(dev)
            try:
                mesh = rsx.Mesh(Q, G)
            except rsx.RsxError as e:            # (collective outcome: every rank raises the same list)
                fails.append((loop, rows, d, "setup", str(e)))
                continue
            meshes += 1
            retries += mesh.export_retries()
            mesh.set_wait_limit(20.0)
            half = rows // 2
            if half > 0:
                mesh.exchange_apply(0, half, 0.5)
            mesh.exchange_apply(half, rows - half, 0.5)
            err = None
            try:
                mesh.check()
            except rsx.RsxError as e:
                err = str(e)
            want = Q0 - 0.5 * sum(Gs)
            ok = err is None and bool(torch.allclose(Q.cpu(), want, rtol=1e-5, atol=1e-5)) and float(G.abs().max()) == 0.0
            if ok:
                G.copy_(Gs[(rank + 1) % world].to(dev))
                mesh.exchange_apply(0, rows, 0.25)
                try:
                    mesh.check()
                except rsx.RsxError as e:
                    err = str(e)
                want2 = want - 0.25 * sum(Gs)
                ok = err is None and bool(torch.allclose(Q.cpu(), want2, rtol=1e-5, atol=1e-5)) and float(G.abs().max()) == 0.0
            # replicas identical: rank 0's rows against everybody's
            box = [None] * world
            dist.all_gather_object(box, Q.cpu().numpy().tobytes())
            same = all(b == box[0] for b in box)
            if not (ok and same):
                fails.append((loop, rows, d, "exchange", err or ("replicas differ" if ok else "wrong sums")))
            if post_barrier:
                mesh.close()
            else:                                # round 5's teardown
                torch.cuda.synchronize()
                tb = time.time()
                dist.barrier()
                td = time.time()
                mesh._destroy()
                mesh.timings.update(barrier_before=td - tb, destroy=time.time() - td)
            for k, v in mesh.timings.items():
                phases[k] = phases.get(k, 0.0) + v
            if empty_cache:
                del mesh, Q, G
                torch.cuda.empty_cache()
    out[rank] = {"fails": fails, "retries": retries, "meshes": meshes, "seconds": round(time.time() - t0, 2),
                 "phase_seconds": {k: round(v, 3) for k, v in phases.items()}}
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--loops", type=int, default=20)
    ap.add_argument("--no-post-barrier", action="store_true")
    ap.add_argument("--own-memory", action="store_true")
    ap.add_argument("--mesh-memory", action="store_true", help="tables in memory from rsx_mesh_alloc (rsx.mesh_tensor)")
    ap.add_argument("--empty-cache", action="store_true", help="torch.cuda.empty_cache() after every mesh (segments freed and re-allocated: "
                    "new allocations may land where a peer's memory was mapped before)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() + 31 * a.world) % 2000
    status = "ok"
    try:
        mp.spawn(worker, args=(a.world, port, a.loops, not a.no_post_barrier, 2 if a.mesh_memory else int(a.own_memory), out, a.empty_cache),
                 nprocs=a.world, join=True)
    except Exception as e:                       # noqa: BLE001
        status = "crashed: " + repr(e)[:2000]
    res = {"world": a.world, "loops": a.loops, "post_barrier": not a.no_post_barrier, "own_memory": a.own_memory, "mesh_memory": a.mesh_memory, "empty_cache": a.empty_cache,
           "status": status,
           "ranks": {int(k): v for k, v in out.items()}}
    res["failures"] = sum(len(v["fails"]) for v in res["ranks"].values())
    res["retries"] = sum(v["retries"] for v in res["ranks"].values())
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "a") as f:
            f.write(line + "\n")
    sys.exit(0 if status == "ok" and res["failures"] == 0 else 1)


if __name__ == "__main__":
    main()
