"""A pool of PERSISTENT rank processes for the multi-process tests (drop-in for `torch.multiprocessing.spawn(fn, args, nprocs, join=True)`).

Every `mp.spawn` starts `nprocs` fresh interpreters that import torch and bring up HIP: about 3 s each on a fresh box, and the
GPU suite started ~100 of them (round 5: 165 processes, 896 s on the driver's box against 352 s warm).  The pool starts a rank process
once (spawn context: never a fork of a process that has touched the GPU) and hands it one task after another; a task is
`fn(rank, *args)` exactly as mp.spawn calls it (the functions bring their process group up and down themselves).  After ANY failure --
a rank raised, the parent was interrupted (pytest-timeout), a rank died -- every process of the pool is killed and started again on
the next use: a rank left inside a collective never poisons the next test.  RSX_RANKPOOL=0 falls back to mp.spawn.

Side effect worth having: the ranks' torch allocator segments are exported over HIP IPC again and again by successive tests (meshes
built, closed, built over the same pooled segment) -- the re-export pattern of a long-running trainer, which fresh processes never see.
"""
import atexit
import importlib
import multiprocessing
import multiprocessing.connection
import os
import sys
import traceback

_CTX = multiprocessing.get_context("spawn")
_workers = []           # [(process, connection)]


def _serve(conn, paths):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for p in paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    while True:
        try:
            task = conn.recv()
        except (EOFError, OSError):
            return
        if task is None:
            return
        mod, name, rank, args = task
        try:
            fn = getattr(importlib.import_module(mod), name)
            fn(rank, *args)
            if "torch" in sys.modules:
                import torch
                if torch.cuda.is_available() and torch.cuda.is_initialized():
                    # (NO torch.cuda.empty_cache() here: a rank that frees segments its peers had mapped over HIP IPC and gets the same
                    #  addresses back hands out handles the peers resolve to the old memory -- profiles/r06_mesh_stress.txt)
                    torch.cuda.synchronize()
            conn.send(("ok", None))
        except BaseException:                       # noqa: BLE001 -- reported to the parent, which restarts the pool
            try:
                conn.send(("err", traceback.format_exc()))
            except Exception:                       # noqa: BLE001
                pass
            return


def _kill_all():
    global _workers
    for p, c in _workers:
        try:
            c.close()
        except Exception:                           # noqa: BLE001
            pass
        if p.is_alive():
            p.kill()
    for p, _ in _workers:
        p.join(timeout=10)
    _workers = []


atexit.register(_kill_all)


def _ensure(n):
    paths = [p for p in sys.path if p and os.path.isdir(p)]
    while len(_workers) < n:
        parent, child = _CTX.Pipe()
        p = _CTX.Process(target=_serve, args=(child, paths), daemon=True)
        p.start()
        child.close()
        _workers.append((p, parent))


def spawn(fn, args=(), nprocs=1, join=True):
    """run fn(rank, *args) on ranks 0 .. nprocs - 1 and wait; raises RuntimeError with the rank's traceback if one fails"""
    assert join, "the pool only runs joined tasks"
    if os.environ.get("RSX_RANKPOOL", "1") == "0":
        import torch.multiprocessing as mp
        return mp.spawn(fn, args=args, nprocs=nprocs, join=True)
    if any(not p.is_alive() for p, _ in _workers):
        _kill_all()
    _ensure(nprocs)
    ok = False
    try:
        for rank in range(nprocs):
            _workers[rank][1].send((fn.__module__, fn.__qualname__, rank, tuple(args)))
        pending = {rank: _workers[rank][1] for rank in range(nprocs)}
        errors = []
        while pending and not errors:
            ready = multiprocessing.connection.wait(list(pending.values()) + [_workers[r][0].sentinel for r in pending], timeout=1.0)
            for rank, conn in list(pending.items()):
                if conn in ready or conn.poll():
                    try:
                        status, tb = conn.recv()
                    except (EOFError, OSError):
                        status, tb = "err", f"rank {rank} died (exit code {_workers[rank][0].exitcode})"
                    if status != "ok":
                        errors.append(f"-- rank {rank} --\n{tb}")
                    del pending[rank]
                elif not _workers[rank][0].is_alive():
                    errors.append(f"-- rank {rank} --\ndied (exit code {_workers[rank][0].exitcode})")
                    del pending[rank]
        if errors:
            raise RuntimeError("a rank of the pool failed:\n" + "\n".join(errors))
        ok = True
    finally:
        if not ok:                                  # a failure, or the parent was interrupted: nobody is left inside a collective
            _kill_all()
