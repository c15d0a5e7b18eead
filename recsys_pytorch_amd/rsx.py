"""ctypes binding of librsx.so (include/rsx.h) for torch tensors.

torch is storage and streams only: every function hands raw device pointers
and torch's current HIP stream to the C ABI.  There is NO fallback: a missing
library or a non-device tensor raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSX_LIB overrides the library path (development: the -DRSX_ABLATE build librsx_dev.so for tools/ablate*.py)
LIB_PATH = os.environ.get("RSX_LIB") or os.path.join(_HERE, "librsx.so")
_lib = None

RSX_USERS_UNIQUE = 1
RSX_NO_UPDATE = 2
RSX_ITEMS_ONLY = 4
RSX_USERS_ONLY = 8
RSX_WIDE_OFFSETS = 16
RSX_DETERMINISTIC = 32
RSX_BATCH_SORTED = 64
RSX_SAMPLE_SORT_POS = 1
RSX_LOSS_SLOTS = 2048
RSX_TRAINER_SLOTS = 3
RSX_MAX_CHUNKS = 8
RSX_PROGRESS_WORDS = 16
RSX_PROGRESS_VIOLATIONS = 8
RSX_COMM_ID_BYTES = 128
RSX_MESH_DESC_BYTES = 512
RSX_EXCHANGE_ALLREDUCE = 1
RSX_EXCHANGE_SCATTER_GATHER = 2
SUPPORTED_DIMS = (32, 64, 128, 256)

# symbol -> (restype, argtypes); mirrors include/rsx.h one to one
_P, _I64, _I32, _F, _U, _U64, _D = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint, C.c_uint64, C.c_double
SIGNATURES = {
    "rsx_version": (C.c_int, []),
    "rsx_last_error": (C.c_char_p, []),
    "rsx_device_info_get": (C.c_int, [C.c_int, _P]),
    "rsx_set_option": (C.c_int, [C.c_char_p, _I64]),
    "rsx_bpr_step_workspace": (_I64, [_I64, _I64, _I32]),
    "rsx_bpr_step_det_workspace": (_I64, [_I64, _I64]),
    "rsx_bpr_step": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, _I64, _I32, _F, _F, _P, _U, _P, _I64,
                               _P, _P, _I32, _I32, _U64, _P]),
    "rsx_fold_hot_grad": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P]),
    "rsx_apply_item_grad": (C.c_int, [_P, _P, _I64, _I32, _F, _P, _P, _I32, _P]),
    "rsx_bpr_grad": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _I64, _I32, _F, _P, _P]),
    "rsx_pointwise_grad": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _I64, _I32, _F, _I32, _P, _P]),
    "rsx_adam_apply": (C.c_int, [_P, _P, _P, _P, _I64, _D, _D, _D, _D, _I64, _P]),
    "rsx_spmm_plan": (_I64, [_P, _I64, _I32, _P, _P, _P]),
    "rsx_spmm_csr": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "rsx_spmm_csr_sparse_rows": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "rsx_spmm_csr_select_rows": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "rsx_spmm_csr_init": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "rsx_spmm_scale_rows": (C.c_int, [_P, _P, _I64, _I32, _F, _P]),
    "rsx_scale": (C.c_int, [_P, _I64, _F, _P]),
    "rsx_spmm_hot_capacity": (_I64, [_I32]),
    "rsx_spmm_hot_chunk_rows": (_I64, [_I32]),
    "rsx_spmm_hot_rows": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "rsx_spmm_mark_batch_rows": (C.c_int, [_P, _I64, _P, _P, _P, _I64, _I64, _P]),
    "rsx_pair_score": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P]),
    "rsx_eval_holdout": (C.c_int, [_I64, _P, _I32, _P, _I32, _P, _P, _P]),
    "rsx_eval_loo": (C.c_int, [_I64, _P, _I32, _P, _I32, _P, _P]),
    "rsx_bpr_sample_workspace": (_I64, [_I64, _I64]),
    "rsx_bpr_sample": (C.c_int, [_P, _P, _I64, _I64, _I64, _U64, _U64, _I64, _I32, _U64, _U, _P, _I64,
                                 _P, _P, _P, _P, _P, _P]),
    "rsx_bpr_build_signature": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "rsx_bpr_item_cdf_workspace": (_I64, [_I64]),
    "rsx_bpr_build_item_cdf": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _I64, _P]),
    "rsx_bpr_trainer_create": (C.c_int, [_P, _P]),
    "rsx_bpr_trainer_destroy": (None, [_P]),
    "rsx_bpr_trainer_run": (C.c_int, [_P, _I64, _I64, _I64, _I32, _P]),
    "rsx_bpr_trainer_state": (C.c_int, [_P, _P, _P]),
    "rsx_bpr_trainer_seek": (C.c_int, [_P, _I64, _I64, _P]),
    "rsx_bpr_trainer_last_batch": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "rsx_bpr_trainer_kernel_ms": (C.c_int, [_P, _P, _P]),
    "rsx_bpr_trainer_check": (C.c_int, [_P, _P]),
    "rsx_chunk_rows": (_I64, [_I64, _I32, _I32]),
    "rsx_bpr_csc_bytes": (_I64, [_I64, _I64]),
    "rsx_bpr_csc_workspace": (_I64, [_I64, _I64]),
    "rsx_bpr_build_csc": (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _I64, _P, _I64, _P, _P]),
    "rsx_csc_destroy": (None, [_P]),
    "rsx_csc_info": (C.c_int, [_P, _P, _P, _P, _P]),
    "rsx_bpr_sample_csc_workspace": (_I64, [_I64]),
    "rsx_bpr_sample_csc": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I32, _U64, _U64, _I32, _U64, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "rsx_bpr_sample_chunked": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _I64, _U64, _U64, _I64, _I32, _U64, _P, _I64,
                                         _P, _P, _P, _P, _P, _P, _P]),
    "rsx_bpr_step_chunked": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I32, _P, _P, _P, _I64, _I32, _F, _F, _P, _P, _P, _I32,
                                       _I32, _U64, _P, _P, _I32, _I32, _P]),
    "rsx_comm_unique_id": (C.c_int, [_P]),
    "rsx_comm_create": (C.c_int, [_P, _I32, _I32, _P]),
    "rsx_comm_destroy": (None, [_P]),
    "rsx_comm_info": (C.c_int, [_P, _P, _P]),
    "rsx_comm_all_reduce_f32": (C.c_int, [_P, _P, _I64, _P]),
    "rsx_mesh_alloc": (C.c_int, [_I64, _P]),
    "rsx_mesh_free": (C.c_int, [_P]),
    "rsx_mesh_alloc_refused": (C.c_int, []),
    "rsx_mesh_local": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "rsx_mesh_connect": (C.c_int, [_P, _I32, _I32, _P]),
    "rsx_mesh_exchange_apply": (C.c_int, [_P, _I64, _I64, _F, _P]),
    "rsx_mesh_set_wait_limit": (C.c_int, [_P, C.c_double]),
    "rsx_mesh_info": (C.c_int, [_P, _P, _P, _P]),
    "rsx_mesh_check": (C.c_int, [_P, _P]),
    "rsx_mesh_export_retries": (C.c_int, [_P]),
    "rsx_mesh_destroy": (None, [_P]),
    "rsx_score": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _P, _P, _P, _P]),
    "rsx_topk": (C.c_int, [_P, _I64, _I64, _I32, _P, _P, _P]),
    "rsx_score_topk_workspace": (_I64, [_I64, _I64]),
    "rsx_score_topk_workspace_d": (_I64, [_I64, _I64, _I32]),
    "rsx_score_topk": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _P, _P, _I32, _P, _P, _P, _I64, _P]),
}


class RsxError(RuntimeError):
    pass


class DeviceInfo(C.Structure):
    _fields_ = [("device", C.c_int), ("compute_units", C.c_int), ("wavefront_size", C.c_int),
                ("total_mem_bytes", C.c_int64), ("lds_bytes_per_cu", C.c_int), ("clock_khz", C.c_int),
                ("arch", C.c_char * 64)]


def lib():
    """Load librsx.so.  Raises if it has not been built (python -m recsys_pytorch_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsxError(f"{LIB_PATH} not found: the HIP extension is not built "
                           "(run `python -m recsys_pytorch_amd.build`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RsxError(f"{what} failed ({rc}): {lib().rsx_last_error().decode()}")


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RsxError(f"{name} must be a device (HIP) tensor; the HIP path has no CPU fallback")
    if t.dtype != dtype or not t.is_contiguous():
        raise RsxError(f"{name} must be contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def version():
    return lib().rsx_version()


def set_option(name, value):
    """include/rsx.h:rsx_set_option ("score_lanes", "sample_sort_cap", "step_waves", "apply_stream", "touched_apply", "mesh_blocks")"""
    _check(lib().rsx_set_option(name.encode(), int(value)), "rsx_set_option")


def device_info(device=0):
    info = DeviceInfo()
    _check(lib().rsx_device_info_get(device, C.byref(info)), "rsx_device_info_get")
    return {f: (getattr(info, f).decode() if f == "arch" else getattr(info, f)) for f, _ in DeviceInfo._fields_}


def bpr_step_workspace(num_users, max_batch, d):
    n = lib().rsx_bpr_step_workspace(num_users, max_batch, d)
    if n < 0:
        raise RsxError("rsx_bpr_step_workspace: invalid shape")
    return n


def bpr_step_det_workspace(batch, num_items):
    n = lib().rsx_bpr_step_det_workspace(batch, num_items)
    if n < 0:
        raise RsxError("rsx_bpr_step_det_workspace: invalid shape")
    return n


class HotItems:
    """Replicated gradient rows for the most popular items (include/rsx.h: hot_slot_dev)."""

    def __init__(self, item_counts, num_hot, replicas, d, device):
        counts = torch.as_tensor(item_counts).to(device)
        num_hot = int(min(num_hot, counts.numel()))
        if not counts.dtype.is_floating_point and counts.numel() and int(counts.max()) < (1 << 31) and int(counts.min()) >= 0:
            # (integer counts tie often: the lower item id wins, so that the same rows are replicated in every launch)
            n = counts.numel()
            key = counts.to(torch.int64) * n + (n - 1 - torch.arange(n, device=counts.device))
            self.items = torch.topk(key, num_hot).indices.to(torch.int32).contiguous()
        else:
            self.items = torch.topk(counts, num_hot).indices.to(torch.int32).contiguous()
        self.slot = torch.full((counts.numel(),), -1, dtype=torch.int32, device=device)
        self.slot[self.items.long()] = torch.arange(num_hot, dtype=torch.int32, device=device)
        self.replicas = int(replicas)
        self.ghot = torch.zeros(num_hot * self.replicas * d, dtype=torch.float32, device=device)
        self.n = num_hot


def fold_hot_grad(G, hot):
    _check(lib().rsx_fold_hot_grad(_dev(G, torch.float32, "G"), _dev(hot.ghot, torch.float32, "ghot"),
                                   _dev(hot.items, torch.int32, "hot items"), hot.n, hot.replicas,
                                   G.shape[1], _stream()), "rsx_fold_hot_grad")


def bpr_step(P, Q, G, u, i, j, lr, inv_batch, loss_acc=None, users_unique=False, ws=None,
             no_update=False, hot=None, neg_block=0, neg_key=0, only=None, wide_offsets=False, deterministic=False, batch_sorted=False):
    """One batch of include/rsx.h:rsx_bpr_step.  u, i, j: int32 device tensors.
    only = "items" | "users": one pass of the two-pass step (RSX_ITEMS_ONLY / RSX_USERS_ONLY)."""
    d = P.shape[1]
    _check(lib().rsx_bpr_step(
        _dev(P, torch.float32, "P"), _dev(Q, torch.float32, "Q"),
        _dev(G, torch.float32, "G") if G is not None else None,
        P.shape[0], Q.shape[0], _dev(u, torch.int32, "u"), _dev(i, torch.int32, "i"),
        _dev(j, torch.int32, "j"), u.numel(), d, float(lr), float(inv_batch),
        _dev(loss_acc, torch.float32, "loss_acc") if loss_acc is not None else None,
        (RSX_USERS_UNIQUE if users_unique else 0) | (RSX_NO_UPDATE if no_update else 0)
        | {None: 0, "items": RSX_ITEMS_ONLY, "users": RSX_USERS_ONLY}[only] | (RSX_WIDE_OFFSETS if wide_offsets else 0)
        | (RSX_DETERMINISTIC if deterministic else 0) | (RSX_BATCH_SORTED if batch_sorted else 0),
        C.c_void_p(ws.data_ptr()) if ws is not None else None,
        ws.numel() * ws.element_size() if ws is not None else 0,
        _dev(hot.slot, torch.int32, "hot slot") if hot is not None else None,
        _dev(hot.ghot, torch.float32, "ghot") if hot is not None else None,
        hot.replicas if hot is not None else 0, int(neg_block), int(neg_key) & (2**64 - 1), _stream()),
        "rsx_bpr_step")


def apply_item_grad(Q, G, lr, hot=None):
    """Q -= lr*G; G = 0.  With `hot` the replicas of the popular rows are folded in here."""
    _check(lib().rsx_apply_item_grad(_dev(Q, torch.float32, "Q"), _dev(G, torch.float32, "G"),
                                     Q.shape[0], Q.shape[1], float(lr),
                                     _dev(hot.slot, torch.int32, "hot slot") if hot is not None else None,
                                     _dev(hot.ghot, torch.float32, "ghot") if hot is not None else None,
                                     hot.replicas if hot is not None else 0, _stream()), "rsx_apply_item_grad")


def bpr_grad(P, Q, GP, GQ, u, i, j, inv_batch, loss_acc=None):
    """dense gradients of one batch (loss.backward(), models/MF.py:67); tables untouched"""
    _check(lib().rsx_bpr_grad(
        _dev(P, torch.float32, "P"), _dev(Q, torch.float32, "Q"), _dev(GP, torch.float32, "GP"),
        _dev(GQ, torch.float32, "GQ"), P.shape[0], Q.shape[0], _dev(u, torch.int32, "u"),
        _dev(i, torch.int32, "i"), _dev(j, torch.int32, "j"), u.numel(), P.shape[1], float(inv_batch),
        _dev(loss_acc, torch.float32, "loss_acc") if loss_acc is not None else None, _stream()), "rsx_bpr_grad")


def pointwise_grad(P, Q, GP, GQ, u, i, y, inv_n, loss_func="ce", loss_acc=None):
    """dense gradients of one POINTWISE batch (models/MF.py:99-102, hparams['pointwise']); loss_func "ce"
    (binary_cross_entropy_with_logits) or "mse"; tables untouched.  GP = GQ = None: the loss alone (into loss_acc)"""
    _check(lib().rsx_pointwise_grad(
        _dev(P, torch.float32, "P"), _dev(Q, torch.float32, "Q"), _dev(GP, torch.float32, "GP") if GP is not None else None,
        _dev(GQ, torch.float32, "GQ") if GQ is not None else None, P.shape[0], Q.shape[0], _dev(u, torch.int32, "u"), _dev(i, torch.int32, "i"),
        _dev(y, torch.float32, "y"), u.numel(), P.shape[1], float(inv_n), int(loss_func == "mse"),
        _dev(loss_acc, torch.float32, "loss_acc") if loss_acc is not None else None, _stream()), "rsx_pointwise_grad")


def adam_apply(W, M, V, G, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update over a whole table (models/MF.py:30), then G = 0"""
    _check(lib().rsx_adam_apply(_dev(W, torch.float32, "W"), _dev(M, torch.float32, "M"),
                                _dev(V, torch.float32, "V"), _dev(G, torch.float32, "G"), W.numel(),
                                float(lr), float(beta1), float(beta2), float(eps), int(t), _stream()),
           "rsx_adam_apply")


class SpmmHot(C.Structure):
    """include/rsx.h: rsx_spmm_hot, field for field"""
    _fields_ = [("num_slots", C.c_int32), ("chunk_rows", C.c_int32), ("num_uniq", C.c_int32), ("reserved", C.c_int32), ("num_src", _I64),
                ("hot_rows", _P), ("uniq_rows", _P), ("src_rows", _P), ("cw_ptr", _P), ("rw_off", _P), ("ent_code", _P), ("ent_val", _P)]


def spmm_hot_plan(csr, d, waves=16):
    """host plan of include/rsx.h: rsx_spmm_hot for the longest rows of `csr` at row width d: (dict of numpy arrays, share of the non-zeros
    it covers), or (None, 0.0) for an empty matrix.  The longest rows get the rsx_spmm_hot_capacity(d) slots; a row that would load one
    wavefront with more than ~1.5 average slots' worth of entries gets several slots (its entries dealt round); the slots are dealt to
    the 16 wavefronts longest first, to the least loaded wavefront that still has a free slot."""
    import numpy as np
    H, K = int(lib().rsx_spmm_hot_capacity(int(d))), int(lib().rsx_spmm_hot_chunk_rows(int(d)))
    R = H // waves
    indptr = np.asarray(csr.indptr, dtype=np.int64)
    lens = np.diff(indptr)
    cand = np.argsort(-lens, kind="stable")[:H]
    cand = cand[lens[cand] > 0]
    if len(cand) == 0:
        return None, 0.0
    cap = max(1.0, 1.5 * float(lens[cand].sum()) / H)
    items, reps, used = [], [], 0
    for it in cand:                                     # (H <= 512: a Python loop)
        n = min(waves, max(1, int(np.ceil(lens[it] / cap))))
        if used + n > H:
            n = H - used if not items else 0
            if n <= 0:
                break
        items.append(int(it)); reps.append(n); used += n
    items, reps = np.asarray(items, np.int64), np.asarray(reps, np.int64)
    # slots -> wavefronts (longest processing time first, at most R slots per wavefront)
    slot_item = np.repeat(np.arange(len(items)), reps)
    slot_rep = np.concatenate([np.arange(n) for n in reps])
    slot_load = (lens[items] / reps)[slot_item]
    wave_load, wave_used = np.zeros(waves), np.zeros(waves, np.int64)
    slot_id = np.empty(len(slot_item), np.int64)
    for q in np.argsort(-slot_load, kind="stable"):
        free = np.flatnonzero(wave_used < R)
        w = free[np.argmin(wave_load[free])]
        slot_id[q] = w * R + wave_used[w]
        wave_used[w] += 1; wave_load[w] += slot_load[q]
    hot_rows = np.full(H, -1, np.int32)
    hot_rows[slot_id] = items[slot_item]
    first_slot = np.concatenate([[0], np.cumsum(reps)[:-1]])                # index into slot_id of every item's replica 0
    sub = csr[items].tocsr()                                                # [items x N], columns ascending inside a row
    sub.sort_indices()
    row_of = np.repeat(np.arange(len(items)), np.diff(sub.indptr))
    rank = np.arange(sub.nnz) - sub.indptr[row_of]                          # position of the entry inside its hot row
    slot = slot_id[first_slot[row_of] + rank % reps[row_of]]
    src_rows = np.unique(sub.indices).astype(np.int32)
    pos = np.searchsorted(src_rows, sub.indices)
    key = ((pos // K) * waves + slot // R) * R + slot % R                  # (chunk, wavefront, slot inside the wavefront)
    order = np.argsort(key, kind="stable")
    nchunks = -(-len(src_rows) // K)
    per_slot = np.bincount(key, minlength=nchunks * waves * R).reshape(nchunks * waves, R)
    cw_ptr = np.concatenate([[0], np.cumsum(per_slot.sum(1))]).astype(np.int64)
    rw_off = np.concatenate([np.zeros((nchunks * waves, 1), np.int64), np.cumsum(per_slot, 1)], 1)
    assert rw_off.max(initial=0) < 65536
    code = ((pos % K) | ((slot % R) << 8)).astype(np.uint16)[order]
    plan = {"num_slots": H, "chunk_rows": K, "hot_rows": hot_rows, "uniq_rows": np.sort(items).astype(np.int32), "src_rows": src_rows,
            "cw_ptr": cw_ptr, "rw_off": rw_off.astype(np.uint16).ravel(), "ent_code": code, "ent_val": sub.data.astype(np.float32)[order]}
    return plan, float(lens[items].sum()) / max(1, csr.nnz)


class SpmmGraph:
    """device CSR of a (square) sparse matrix + its segment plan (include/rsx.h:rsx_spmm_plan).  d (the row width the products will
    run at) lets the LONGEST rows be computed by scatter (include/rsx.h: rsx_spmm_hot_rows) when they hold a worthwhile share of the
    non-zeros -- hot = None: when the rsx_spmm_hot_capacity(d) longest rows hold at least 10 % of at least a million non-zeros;
    True / False: always / never (RSX_SPMM_HOT=1 / 0 overrides the default)"""

    def __init__(self, csr, device, max_seg=None, d=None, hot=None):
        import numpy as np
        if max_seg is None:
            # rows above max_seg non-zeros are cut and combined with atomics.  Measured at the configs[4] shape (ms per product /
            # per training step): 64: 4.39 / 29.1, 128: 3.35 / 21.6, 256: 2.88 / 19.3, 512: 2.67 / 17.2, 1024: 2.58 / 16.8,
            # 2048: 2.60 / 17.0, 4096: 2.66 / 17.7, 16384: 4.8 / 31 (one lane group then walks a segment for milliseconds)
            max_seg = int(os.environ.get("RSX_SPMM_MAX_SEG", 1024))
        csr = csr.tocsr()
        csr.sort_indices()
        indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        n = csr.shape[0]
        cnt = lib().rsx_spmm_plan(indptr.ctypes.data, n, max_seg, None, None, None)
        if cnt < 0:
            raise RsxError("rsx_spmm_plan failed")
        row, beg, ln = np.empty(cnt, np.int32), np.empty(cnt, np.int64), np.empty(cnt, np.int32)
        lib().rsx_spmm_plan(indptr.ctypes.data, n, max_seg, row.ctypes.data, beg.ctypes.data, ln.ctypes.data)
        # the longest rows by scatter: taken out of the plan, described by source row (include/rsx.h: rsx_spmm_hot)
        self.hot = None
        if hot is None and os.environ.get("RSX_SPMM_HOT") in ("0", "1"):
            hot = os.environ["RSX_SPMM_HOT"] == "1"
        if d is not None and hot is not False and csr.nnz > 0:
            plan, share = spmm_hot_plan(csr, d) if (hot is True or csr.nnz >= 1_000_000) else (None, 0.0)
            if plan is not None and (hot is True or share >= 0.10):
                dev = lambda a, view=None: torch.from_numpy(a if view is None else a.view(view)).to(device).contiguous()
                self._hot_keep = {k: dev(plan[k], np.int16 if k in ("ent_code", "rw_off") else None)
                                  for k in ("hot_rows", "uniq_rows", "src_rows", "cw_ptr", "rw_off", "ent_code", "ent_val")}
                self.hot = SpmmHot(num_slots=plan["num_slots"], chunk_rows=plan["chunk_rows"], num_uniq=len(plan["uniq_rows"]), reserved=0,
                                   num_src=len(plan["src_rows"]), **{k: v.data_ptr() for k, v in self._hot_keep.items()})
                self.hot_share = share
                keep = ~np.isin(row, plan["uniq_rows"])   # the plan owns no segment for them
                row, beg, ln = row[keep], beg[keep], ln[keep]
                cnt = int(keep.sum())
        # longest segments first (the long item rows of a popularity-skewed graph would otherwise start last and finish alone)
        order = np.argsort(-ln.astype(np.int64), kind="stable")
        row, beg, ln = row[order].copy(), beg[order].copy(), ln[order].copy()
        to = lambda a: torch.from_numpy(a).to(device).contiguous()
        self.n, self.num_segs = n, int(cnt)
        self.seg_row, self.seg_begin, self.seg_len = to(row), to(beg), to(ln)
        self.indptr = to(indptr)
        self.indices = to(np.ascontiguousarray(csr.indices, dtype=np.int32))
        self.vals = to(np.ascontiguousarray(csr.data, dtype=np.float32))


def spmm(graph, X, Y, S_acc=None, x_nonzero=None, y_wanted=None, S_init=None):
    """Y = A X (and S_acc += A X) -- include/rsx.h:rsx_spmm_csr; x_nonzero (uint8 [N], 0 = that row of X is entirely zero):
    rsx_spmm_csr_sparse_rows, which does not fetch such rows; y_wanted (uint8 [N]): rsx_spmm_csr_select_rows, which computes
    only the flagged rows of Y / S_acc (the others keep what they held); S_init: rsx_spmm_csr_init, S_acc = S_init + A X (S_acc is
    overwritten: the first product of a propagation, without the copy of the source into the running sum)"""
    if (x_nonzero is not None or S_init is not None) and y_wanted is not None:
        raise RsxError("spmm: y_wanted does not combine with x_nonzero / S_init (different products)")
    if graph.num_segs > 0:                               # (a small graph whose every non-empty row goes by scatter has no segment left)
        _spmm_planned(graph, X, Y, S_acc, x_nonzero, y_wanted, S_init)
    if getattr(graph, "hot", None) is not None:          # the longest rows, which the plan leaves out: by scatter
        ptr = lambda t, dt, name: _dev(t, dt, name) if t is not None else None
        _check(lib().rsx_spmm_hot_rows(C.byref(graph.hot), _dev(X, torch.float32, "X"), ptr(x_nonzero, torch.uint8, "x_nonzero"),
                                       ptr(y_wanted, torch.uint8, "y_wanted"), ptr(S_init, torch.float32, "S_init"), _dev(Y, torch.float32, "Y"),
                                       ptr(S_acc, torch.float32, "S_acc"), graph.n, X.shape[1], _stream()), "rsx_spmm_hot_rows")


def _spmm_planned(graph, X, Y, S_acc, x_nonzero, y_wanted, S_init):
    if S_init is not None:
        _check(lib().rsx_spmm_csr_init(
            _dev(graph.seg_row, torch.int32, "seg_row"), _dev(graph.seg_begin, torch.int64, "seg_begin"),
            _dev(graph.seg_len, torch.int32, "seg_len"), graph.num_segs, _dev(graph.indptr, torch.int64, "indptr"),
            _dev(graph.indices, torch.int32, "indices"), _dev(graph.vals, torch.float32, "vals"), _dev(X, torch.float32, "X"),
            _dev(x_nonzero, torch.uint8, "x_nonzero") if x_nonzero is not None else None, _dev(S_init, torch.float32, "S_init"),
            _dev(Y, torch.float32, "Y"), _dev(S_acc, torch.float32, "S_acc"), graph.n, X.shape[1], _stream()), "rsx_spmm_csr_init")
        return
    common = (_dev(graph.seg_row, torch.int32, "seg_row"), _dev(graph.seg_begin, torch.int64, "seg_begin"),
              _dev(graph.seg_len, torch.int32, "seg_len"), graph.num_segs, _dev(graph.indptr, torch.int64, "indptr"),
              _dev(graph.indices, torch.int32, "indices"), _dev(graph.vals, torch.float32, "vals"), _dev(X, torch.float32, "X"))
    tail = (_dev(Y, torch.float32, "Y"), _dev(S_acc, torch.float32, "S_acc") if S_acc is not None else None, graph.n, X.shape[1],
            _stream())
    if y_wanted is not None:
        _check(lib().rsx_spmm_csr_select_rows(*common, _dev(y_wanted, torch.uint8, "y_wanted"), *tail), "rsx_spmm_csr_select_rows")
    elif x_nonzero is None:
        _check(lib().rsx_spmm_csr(*common, *tail), "rsx_spmm_csr")
    else:
        _check(lib().rsx_spmm_csr_sparse_rows(*common, _dev(x_nonzero, torch.uint8, "x_nonzero"), *tail), "rsx_spmm_csr_sparse_rows")


def mark_batch_rows(flags, u, i, j, item_offset):
    """include/rsx.h:rsx_spmm_mark_batch_rows -- flags (uint8 [N]) = 1 exactly at the rows a batch's gradient touches"""
    _check(lib().rsx_spmm_mark_batch_rows(_dev(flags, torch.uint8, "flags"), flags.numel(), _dev(u, torch.int32, "u"),
                                          _dev(i, torch.int32, "i"), _dev(j, torch.int32, "j"), u.numel(), int(item_offset),
                                          _stream()), "rsx_spmm_mark_batch_rows")


def scale_rows(X, flags, alpha):
    """include/rsx.h:rsx_spmm_scale_rows -- X[row] *= alpha where flags[row] != 0 (alpha = 0: the rows are cleared)"""
    _check(lib().rsx_spmm_scale_rows(_dev(X, torch.float32, "X"), _dev(flags, torch.uint8, "flags"), X.shape[0], X.shape[1],
                                     float(alpha), _stream()), "rsx_spmm_scale_rows")


def scale(X, alpha):
    _check(lib().rsx_scale(_dev(X, torch.float32, "X"), X.numel(), float(alpha), _stream()), "rsx_scale")


def pair_score(P, Q, u, i):
    """r[b] = <P[u[b]], Q[i[b]]>  (MF.forward, models/MF.py:38-42)"""
    out = torch.empty(u.numel(), dtype=torch.float32, device=P.device)
    _check(lib().rsx_pair_score(_dev(P, torch.float32, "P"), _dev(Q, torch.float32, "Q"),
                                _dev(u, torch.int32, "u"), _dev(i, torch.int32, "i"), u.numel(),
                                P.shape[1], _dev(out, torch.float32, "out"), _stream()), "rsx_pair_score")
    return out


def eval_holdout(rankings, ks, truth_indptr, truth_indices):
    """HOST function: numpy in, numpy out (include/rsx.h:rsx_eval_holdout)."""
    import numpy as np
    rankings = np.ascontiguousarray(rankings, dtype=np.int32)
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    tp = np.ascontiguousarray(truth_indptr, dtype=np.int64)
    ti = np.ascontiguousarray(truth_indices, dtype=np.int32)
    n, max_k = rankings.shape
    if len(tp) != n + 1:
        raise RsxError("truth_indptr must have one entry per ranked user plus one")
    res = np.zeros((n, 3 * len(ks)), dtype=np.float32)
    _check(lib().rsx_eval_holdout(n, rankings.ctypes.data, max_k, ks.ctypes.data, len(ks),
                                  tp.ctypes.data, ti.ctypes.data, res.ctypes.data), "rsx_eval_holdout")
    return res


def eval_loo(rankings, ks, truth):
    """HOST function: numpy in, numpy out (include/rsx.h:rsx_eval_loo); truth = one held-out item per ranked user"""
    import numpy as np
    rankings = np.ascontiguousarray(rankings, dtype=np.int32)
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    truth = np.ascontiguousarray(truth, dtype=np.int32)
    n, max_k = rankings.shape
    if len(truth) != n:
        raise RsxError("truth must hold one item per ranked user")
    res = np.zeros((n, 2 * len(ks)), dtype=np.float32)
    _check(lib().rsx_eval_loo(n, rankings.ctypes.data, max_k, ks.ctypes.data, len(ks), truth.ctypes.data,
                              res.ctypes.data), "rsx_eval_loo")
    return res


def build_signature(indptr, indices, neg_block):
    """per-user block signature for the sampler's rejection test (include/rsx.h)"""
    sig = torch.empty((indptr.numel() - 1, 2), dtype=torch.int64, device=indptr.device)   # (bits, row start | length << 40)
    _check(lib().rsx_bpr_build_signature(_dev(indptr, torch.int64, "indptr"), _dev(indices, torch.int32, "indices"),
                                         indptr.numel() - 1, int(neg_block), _dev(sig, torch.int64, "sig"),
                                         _stream()), "rsx_bpr_build_signature")
    return sig


def build_item_cdf(indptr, indices, num_items):
    """CDF of the positive-item distribution, lets the sampler order a batch without a device sort"""
    cdf = torch.empty(num_items + 1, dtype=torch.int32, device=indptr.device)
    ws = torch.empty(lib().rsx_bpr_item_cdf_workspace(num_items), dtype=torch.uint8, device=indptr.device)
    _check(lib().rsx_bpr_build_item_cdf(_dev(indptr, torch.int64, "indptr"), _dev(indices, torch.int32, "indices"),
                                        indptr.numel() - 1, num_items, _dev(cdf, torch.int32, "cdf"),
                                        C.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "rsx_bpr_build_item_cdf")
    return cdf


def bpr_sample_workspace(batch, num_items):
    n = lib().rsx_bpr_sample_workspace(batch, num_items)
    if n < 0:
        raise RsxError("rsx_bpr_sample_workspace: invalid shape")
    return n


def bpr_sample(indptr, indices, num_items, batch, seed, step, epoch_pos, u_out, i_out, j_out, neg_block=0,
               neg_key=0, sort_pos=False, ws=None, user_sig=None, item_cdf=None):
    _check(lib().rsx_bpr_sample(
        _dev(indptr, torch.int64, "indptr"), _dev(indices, torch.int32, "indices"),
        indptr.numel() - 1, num_items, batch, seed & (2**64 - 1), step, epoch_pos, int(neg_block),
        int(neg_key) & (2**64 - 1), RSX_SAMPLE_SORT_POS if sort_pos else 0,
        C.c_void_p(ws.data_ptr()) if ws is not None else None,
        ws.numel() * ws.element_size() if ws is not None else 0,
        _dev(user_sig, torch.int64, "user_sig") if user_sig is not None else None,
        _dev(item_cdf, torch.int32, "item_cdf") if item_cdf is not None else None,
        _dev(u_out, torch.int32, "u_out"), _dev(i_out, torch.int32, "i_out"),
        _dev(j_out, torch.int32, "j_out"), _stream()), "rsx_bpr_sample")


class Csc:
    """The transposed interaction matrix of one CSR for the whole-pass sampler (include/rsx.h: rsx_bpr_build_csc /
    rsx_bpr_sample_csc): entries item by item, users ascending, 6 (or 8) bytes each.  Keeps the CSR tensors it was built from and
    its own device blob alive."""

    def __init__(self, indptr, indices, num_items):
        U, nnz = indptr.numel() - 1, indices.numel()
        nb, nw = lib().rsx_bpr_csc_bytes(nnz, int(num_items)), lib().rsx_bpr_csc_workspace(nnz, int(num_items))
        if nb < 0 or nw < 0:
            raise RsxError("rsx_bpr_csc_bytes: invalid shape")
        self.blob = torch.empty(nb, dtype=torch.uint8, device=indptr.device)
        ws = torch.empty(nw, dtype=torch.uint8, device=indptr.device)
        self._keep = (indptr, indices)
        self.num_items, self.nnz = int(num_items), nnz
        self._h = C.c_void_p()
        _check(lib().rsx_bpr_build_csc(_dev(indptr, torch.int64, "indptr"), _dev(indices, torch.int32, "indices") if nnz else None, U, int(num_items), nnz,
                                       C.c_void_p(self.blob.data_ptr()), nb, C.c_void_p(ws.data_ptr()), nw, _stream(), C.byref(self._h)),
               "rsx_bpr_build_csc")
        torch.cuda.current_stream().synchronize()       # (the build scratch is released below)
        self.sample_ws_bytes = int(lib().rsx_bpr_sample_csc_workspace(nnz))

    @property
    def handle(self):
        return self._h

    def info(self):
        """{nnz, num_items, entry_bytes, tiles}"""
        nnz, ni, eb, tiles = C.c_int64(), C.c_int64(), C.c_int(), C.c_int64()
        _check(lib().rsx_csc_info(self._h, C.byref(nnz), C.byref(ni), C.byref(eb), C.byref(tiles)), "rsx_csc_info")
        return {"nnz": nnz.value, "num_items": ni.value, "entry_bytes": eb.value, "tiles": tiles.value}

    def arrays(self):
        """{ptr int64 [I + 1], tile_item int32 [tiles + 1], users uint32 [nnz], rank, deg, tile: entries per tile} on the host -- tests"""
        import numpy as np
        inf = self.info()
        a256 = lambda x: (x + 255) // 256 * 256
        tiles, I = inf["tiles"], inf["num_items"]
        # (the layout mirrors csc_layout in csrc/rsx_sample.hip)
        raw = self.blob.cpu().numpy()
        off_tile = a256((I + 1) * 8)
        off_users = off_tile + a256((tiles + 1) * 4)
        nnz_pad = (self.blob.numel() - off_users) // 8
        off_rd = off_users + a256(nnz_pad * 4)
        ptr = raw[:(I + 1) * 8].view(np.int64)
        tile_item = raw[off_tile:off_tile + (tiles + 1) * 4].view(np.int32)
        users = raw[off_users:off_users + self.nnz * 4].view(np.uint32)
        if inf["entry_bytes"] == 6:
            rd = raw[off_rd:off_rd + self.nnz * 2].view(np.uint16).astype(np.uint32)
            rank, deg = rd & 0xFF, rd >> 8
        else:
            rd = raw[off_rd:off_rd + self.nnz * 4].view(np.uint32)
            rank, deg = rd & 0xFFFF, rd >> 16
        return {"ptr": ptr.copy(), "tile_item": tile_item.copy(), "users": users.copy(), "rank": rank, "deg": deg, "tile": nnz_pad // tiles}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rsx_csc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                              # noqa: BLE001 -- interpreter shutdown
            pass


def bpr_sample_csc(csc, indptr, indices, num_items, seed, step, u_out, i_out, j_out, neg_block=0, neg_key=0, ws=None, user_sig=None,
                   chunks=1, items_real=None, chunk_pos=None):
    """include/rsx.h:rsx_bpr_sample_csc -- a batch of EVERY user once, ordered by positive item by one walk over the CSC"""
    _check(lib().rsx_bpr_sample_csc(
        csc.handle, _dev(indptr, torch.int64, "indptr"), _dev(indices, torch.int32, "indices") if indices.numel() else None, indptr.numel() - 1, int(num_items),
        int(items_real if items_real is not None else num_items), int(chunks), seed & (2**64 - 1), step, int(neg_block),
        int(neg_key) & (2**64 - 1), C.c_void_p(ws.data_ptr()), ws.numel() * ws.element_size(),
        _dev(user_sig, torch.int64, "user_sig") if user_sig is not None else None,
        _dev(u_out, torch.int32, "u_out"), _dev(i_out, torch.int32, "i_out"), _dev(j_out, torch.int32, "j_out"),
        _dev(chunk_pos, torch.int64, "chunk_pos") if chunk_pos is not None else None, _stream()), "rsx_bpr_sample_csc")


def csc_positive_rank(users, deg, seed, step):
    """host restatement (numpy) of the walk's choice of the positive: rank in [0, deg) of every user's sampled item
    (csrc/rsx_sample.hip: csc_step_key / csc_hash) -- tests compare the device pairs against it"""
    import numpy as np
    M = (1 << 64) - 1

    def splitmix64(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    k = splitmix64((int(seed) & M) ^ ((int(step) * 0x9E3779B97F4A7C15) & M) ^ 0xC5C0DE5A3D1E7)
    k0, k1 = np.uint32(k & 0xFFFFFFFF), np.uint32(k >> 32)
    x = np.asarray(users).astype(np.uint32) ^ k0
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    x ^= k1
    return ((x.astype(np.uint64) * np.asarray(deg).astype(np.uint64)) >> np.uint64(32)).astype(np.int64)


def chunk_rows(items_real, chunks, neg_block):
    """rows per item range of the chunked step (include/rsx.h: "item chunks")"""
    n = lib().rsx_chunk_rows(int(items_real), int(chunks), int(neg_block))
    if n < 0:
        raise RsxError("rsx_chunk_rows: invalid shape")
    return n


def bpr_sample_chunked(indptr, indices, num_items, items_real, chunks, batch, seed, step, epoch_pos, u_out, i_out, j_out,
                       chunk_pos, neg_block, neg_key, ws, item_cdf, user_sig=None):
    """include/rsx.h:rsx_bpr_sample_chunked (relabelled CSR; chunk_pos: int64 [chunks + 1] device)"""
    _check(lib().rsx_bpr_sample_chunked(
        _dev(indptr, torch.int64, "indptr"), _dev(indices, torch.int32, "indices"), indptr.numel() - 1, int(num_items),
        int(items_real), int(chunks), int(batch), seed & (2**64 - 1), step, epoch_pos, int(neg_block), int(neg_key) & (2**64 - 1),
        C.c_void_p(ws.data_ptr()), ws.numel() * ws.element_size(),
        _dev(user_sig, torch.int64, "user_sig") if user_sig is not None else None, _dev(item_cdf, torch.int32, "item_cdf"),
        _dev(u_out, torch.int32, "u_out"), _dev(i_out, torch.int32, "i_out"), _dev(j_out, torch.int32, "j_out"),
        _dev(chunk_pos, torch.int64, "chunk_pos"), _stream()), "rsx_bpr_sample_chunked")


def bpr_step_chunked(P, Q, G, items_real, chunks, u, i, j, lr, inv_batch, chunk_pos, progress, neg_block, neg_key, loss_acc=None,
                     hot=None, first_range=0, num_ranges=None):
    """include/rsx.h:rsx_bpr_step_chunked over the item ranges [first_range, first_range + num_ranges) (default: all);
    progress: int32 [RSX_PROGRESS_WORDS] device, zero before the first launch of a step"""
    _check(lib().rsx_bpr_step_chunked(
        _dev(P, torch.float32, "P"), _dev(Q, torch.float32, "Q"), _dev(G, torch.float32, "G"), P.shape[0], Q.shape[0],
        int(items_real), int(chunks), _dev(u, torch.int32, "u"), _dev(i, torch.int32, "i"), _dev(j, torch.int32, "j"), u.numel(),
        P.shape[1], float(lr), float(inv_batch), _dev(loss_acc, torch.float32, "loss_acc") if loss_acc is not None else None,
        _dev(hot.slot, torch.int32, "hot slot") if hot is not None else None,
        _dev(hot.ghot, torch.float32, "ghot") if hot is not None else None, hot.replicas if hot is not None else 0,
        int(neg_block), int(neg_key) & (2**64 - 1), _dev(chunk_pos, torch.int64, "chunk_pos"),
        _dev(progress, torch.int32, "progress"), int(first_range), int(chunks - first_range if num_ranges is None else num_ranges),
        _stream()), "rsx_bpr_step_chunked")


class Comm:
    """RCCL communicator owned by the library (include/rsx.h: rsx_comm_*), one per process, on the current device.
    The 128-byte unique id travels through torch.distributed (any backend: host-side bootstrap only)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = C.create_string_buffer(RSX_COMM_ID_BYTES)
        err = None
        if self.rank == 0:
            # a failure here must still reach the broadcast below: the other ranks are waiting in it
            if lib().rsx_comm_unique_id(ident) != 0:
                err = "rsx_comm_unique_id failed: " + lib().rsx_last_error().decode()
        box = [(err, ident.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        err, raw = box[0]
        if err is not None:                            # every rank raises the same error: nobody is left inside a collective
            raise RsxError(err)
        self._h = C.c_void_p()
        _check(lib().rsx_comm_create(C.create_string_buffer(raw, RSX_COMM_ID_BYTES), self.rank, self.world, C.byref(self._h)),
               "rsx_comm_create")

    def info(self):
        """(rank, world) as RCCL's communicator reports them (include/rsx.h: rsx_comm_info)"""
        r, w = C.c_int(-1), C.c_int(-1)
        _check(lib().rsx_comm_info(self._h, C.byref(r), C.byref(w)), "rsx_comm_info")
        return int(r.value), int(w.value)

    @property
    def handle(self):
        return self._h

    def all_reduce(self, t):
        """sum over the ranks, in place, on torch's current stream"""
        _check(lib().rsx_comm_all_reduce_f32(self._h, _dev(t, torch.float32, "buffer"), t.numel(), _stream()), "rsx_comm_all_reduce_f32")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rsx_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                              # noqa: BLE001 -- interpreter shutdown
            pass


class _MeshMemory:
    """one allocation of rsx_mesh_alloc, freed with its last tensor view (include/rsx.h: memory that HAS been exported already)"""

    def __init__(self, nbytes, shape):
        self.ptr = C.c_void_p()
        _check(lib().rsx_mesh_alloc(int(nbytes), C.byref(self.ptr)), "rsx_mesh_alloc")
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<f4", "data": (int(self.ptr.value), False),
                                         "version": 2, "strides": None}

    def __del__(self):
        try:
            if getattr(self, "ptr", None) is not None and self.ptr.value:
                lib().rsx_mesh_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:                              # noqa: BLE001 -- interpreter shutdown
            pass


def mesh_tensor(rows, d, device=None):
    """a zero-filled fp32 [rows x d] device tensor in memory from rsx_mesh_alloc: plain hipMalloc'ed memory that the library has already
    exported for the peers (an allocation the runtime refuses to export never comes back from there) -- where the tables of a
    BPREngine(exchange="direct") live.  Freed when the last view of it is dropped (after Mesh.close() on every rank)."""
    if device is not None and torch.device(device).index is not None:
        assert torch.device(device).index == torch.cuda.current_device(), "mesh memory is allocated on the CURRENT device"
    mem = _MeshMemory(int(rows) * int(d) * 4, (rows, d))
    t = torch.as_tensor(mem, device=torch.device("cuda", torch.cuda.current_device()))
    assert t.data_ptr() == mem.ptr.value and t.dtype == torch.float32 and tuple(t.shape) == (rows, d)
    t._rsx_mesh_memory = mem                            # (torch holds the exporter too; this names it for the reader)
    return t


def mesh_alloc_refused():
    """allocations rsx_mesh_alloc set aside in this process because the runtime would not export them"""
    return int(lib().rsx_mesh_alloc_refused())


class Mesh:
    """The library's own exchange of the item gradients over xGMI (include/rsx.h: rsx_mesh_*): every rank maps the peers' Q, G and
    a mailbox (HIP IPC), sums ITS slice of the rows by reading the peers directly, applies it, and the updated rows are copied
    from their owners -- one process per GPU, the descriptors travel through torch.distributed (any backend: host bootstrap only).
    Q and G [rows x d] must stay allocated on every rank until close(); close() is collective (a barrier first: no peer may still
    be reading this rank's buffers)."""

    def __init__(self, Q, G, group=None):
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert Q.shape == G.shape and Q.dim() == 2
        self._keep = (Q, G)
        desc = C.create_string_buffer(RSX_MESH_DESC_BYTES)
        self._h = C.c_void_p()
        import time
        self.timings = {}                              # seconds per phase of set-up / teardown (tools/mesh_stress.py reports them)
        t0 = time.perf_counter()
        rc = lib().rsx_mesh_local(_dev(Q, torch.float32, "Q"), _dev(G, torch.float32, "G"), Q.shape[0], Q.shape[1], desc, C.byref(self._h))
        err = None if rc == 0 else f"rsx_mesh_local failed ({rc}): {lib().rsx_last_error().decode()}"
        self.timings["export"] = time.perf_counter() - t0
        box = [None] * self.world
        dist.all_gather_object(box, (err, desc.raw), group=group)          # (a failed rank still takes part: nobody is left waiting)
        bad = [f"rank {q}: {e}" for q, (e, _) in enumerate(box) if e is not None]
        if not bad:
            t0 = time.perf_counter()
            rc = lib().rsx_mesh_connect(self._h, self.rank, self.world, C.create_string_buffer(b"".join(raw for _, raw in box), RSX_MESH_DESC_BYTES * self.world))
            self.timings["open"] = time.perf_counter() - t0
            err = None if rc == 0 else f"rsx_mesh_connect failed ({rc}): {lib().rsx_last_error().decode()}"
            box = [None] * self.world
            dist.all_gather_object(box, err, group=group)
            bad = [f"rank {q}: {e}" for q, e in enumerate(box) if e is not None]
        if bad:                                        # (the same list on every rank: all of them leave, together)
            self._destroy()
            dist.barrier(group=group)                  # nobody exports again while a peer still detaches
            raise RsxError("; ".join(bad))

    @property
    def handle(self):
        return self._h

    def exchange_apply(self, first_row, rows, lr):
        """Q[first_row:first_row + rows] -= lr * (sum over the ranks of G[...]) on every rank, those rows of G zeroed; on torch's
        current stream, behind whatever completed G there"""
        _check(lib().rsx_mesh_exchange_apply(self._h, int(first_row), int(rows), float(lr), _stream()), "rsx_mesh_exchange_apply")

    def set_wait_limit(self, seconds):
        _check(lib().rsx_mesh_set_wait_limit(self._h, float(seconds)), "rsx_mesh_set_wait_limit")

    def info(self):
        r, w, n = C.c_int(-1), C.c_int(-1), C.c_int64(-1)
        _check(lib().rsx_mesh_info(self._h, C.byref(r), C.byref(w), C.byref(n)), "rsx_mesh_info")
        return int(r.value), int(w.value), int(n.value)

    def export_retries(self):
        """hipIpcGetMemHandle calls that failed before rsx_mesh_local's exports succeeded (0 normally)"""
        return int(lib().rsx_mesh_export_retries(self._h))

    def check(self):
        """synchronises; raises if a wait for a peer's signal gave up (the rows of that exchange are wrong)"""
        _check(lib().rsx_mesh_check(self._h, _stream()), "rsx_mesh_check")

    def check_all(self):
        """collective form of check(): every rank learns every rank's result BEFORE anyone raises, so a rank whose wait gave up
        does not leave the others inside the next collective"""
        import torch.distributed as dist
        try:
            self.check()
            err = None
        except RsxError as e:
            err = str(e)
        box = [None] * self.world
        dist.all_gather_object(box, err, group=self.group)
        bad = [f"rank {q}: {e}" for q, e in enumerate(box) if e is not None]
        if bad:
            raise RsxError("; ".join(bad))

    def _destroy(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rsx_mesh_destroy(self._h)
            self._h = C.c_void_p()

    def close(self):
        """collective.  Barrier, unmap + free, barrier: nobody reads my buffers any more when I unmap, and nobody exports the same
        allocations again (a later Mesh over the same pooled segment) while a peer is still detaching from this export."""
        if getattr(self, "_h", None) is not None and self._h.value:
            import time
            import torch.distributed as dist
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.barrier(group=self.group)             # every rank has finished every exchange: nobody reads my buffers any more
            t1 = time.perf_counter()
            self._destroy()
            t2 = time.perf_counter()
            dist.barrier(group=self.group)             # every rank has closed every mapping
            self.timings.update(barrier_before=t1 - t0, destroy=t2 - t1, barrier_after=time.perf_counter() - t2)
            self._keep = None

    def __del__(self):
        try:                                           # best effort (not collective): a dropped mesh must not leak mappings + mailbox
            self._destroy()
        except Exception:                              # noqa: BLE001 -- interpreter shutdown
            pass


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
EXCHANGE_RANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p)   # (ctx, range, G rows, n floats, stream)


class TrainerConfig(C.Structure):
    """include/rsx.h:rsx_bpr_trainer_config, field for field"""
    _fields_ = [("P", _P), ("Q", _P), ("G", _P), ("num_users", _I64), ("num_items", _I64), ("d", C.c_int32),
                ("lr", _F), ("indptr", _P), ("indices", _P), ("batch", _I64), ("seed", _U64), ("seed_key", _U64),
                ("neg_block", C.c_int32), ("two_pass", C.c_int32), ("sample_ws", _P), ("sample_ws_bytes", _I64),
                ("user_sig", _P), ("item_cdf", _P), ("triplets", _P), ("hot_slot", _P), ("G_hot", _P),
                ("hot_items", _P), ("n_hot", C.c_int32), ("hot_replicas", C.c_int32), ("loss_acc", _P),
                ("exchange_begin", EXCHANGE_FN), ("exchange_end", EXCHANGE_FN), ("exchange_ctx", _P),
                ("exchange_applies", C.c_int32), ("sort_min_batch", C.c_int32), ("step0", _I64), ("epoch_pos0", _I64),
                ("G_alt", _P), ("stale_exchange", C.c_int32), ("exchange_kind", C.c_int32), ("comm", _P),
                ("item_rows_padded", _I64), ("chunks", C.c_int32), ("reserved0", C.c_int32), ("items_real", _I64),
                ("chunk_pos", _P), ("progress", _P), ("exchange_range", EXCHANGE_RANGE_FN), ("mesh", _P), ("csc", _P), ("touched", _P)]


class BPRTrainer:
    """The native batch loop (include/rsx.h: rsx_bpr_trainer_*): n steps of sampler || step kernel ->
    [exchange] -> apply per call, queued from C++ on torch's current stream.  Keeps every tensor it
    borrows alive."""

    def __init__(self, P, Q, G, indptr, indices, lr, batch, seed, seed_key, neg_block=0, hot=None, user_sig=None,
                 item_cdf=None, loss_acc=None, exchange=None, two_pass=False, exchange_applies=False, sort_min_batch=0, step0=0, epoch_pos0=0,
                 G_alt=None, comm=None, exchange_kind=0, item_rows_padded=0, chunks=0, items_real=0, num_items=None, exchange_range=None,
                 mesh=None, csc=None):
        """exchange: (begin, end) callables (the collective stays with the caller) OR comm: an rsx.Comm (the library issues it).
        csc: an rsx.Csc over this CSR -- whole-pass batches of an ordered layout are then sampled by the CSC walk.
        chunks > 1: P / Q / G / the CSR live in the caller's relabelled item space of chunks * chunk_rows ids; sharded without
        comm: exchange_range(k, first_row, rows, stream_handle) queues the caller's all-reduce of G[first_row:first_row + rows] on
        the given HIP stream (include/rsx.h: exchange_range)."""
        dev = P.device
        self.batch = int(batch)
        self.chunks = int(chunks)
        self.chunk_pos = self.progress = None
        if self.chunks > 1:
            self.chunk_pos = torch.zeros(RSX_TRAINER_SLOTS * (self.chunks + 1), dtype=torch.int64, device=dev)
            self.progress = torch.zeros(RSX_PROGRESS_WORDS, dtype=torch.int32, device=dev)
        self.triplets = torch.empty(RSX_TRAINER_SLOTS * 3 * self.batch, dtype=torch.int32, device=dev)
        self.sample_ws = None
        if neg_block or sort_min_batch or self.chunks > 1:
            self.sample_ws = torch.empty(max(bpr_sample_workspace(self.batch, int(num_items) if num_items is not None else Q.shape[0]),
                                             csc.sample_ws_bytes if csc is not None else 0), dtype=torch.uint8, device=dev)
        else:
            csc = None                                 # (the plain layout never takes the walk)
        # row marks of the small-batch apply (include/rsx.h: touched): unsharded, unchunked trainers
        self.touched = None
        if exchange is None and comm is None and mesh is None and exchange_range is None and self.chunks <= 1:
            self.touched = torch.zeros(int(num_items) if num_items is not None else Q.shape[0], dtype=torch.uint8, device=dev)
        self._keep = (P, Q, G, indptr, indices, hot, user_sig, item_cdf, loss_acc, G_alt, comm, mesh, csc)
        ptr = lambda t, dt, name: _dev(t, dt, name) if t is not None else None
        self._cb = (None, None)
        if exchange is not None:                       # (begin, end) callables; exceptions become error codes
            def wrap(fn):
                def call(_ctx):
                    try:
                        fn()
                        return 0
                    except Exception as e:             # noqa: BLE001 -- must not unwind through the C frames
                        self._exc = e
                        return 1
                return EXCHANGE_FN(call)
            self._cb = (wrap(exchange[0]), wrap(exchange[1]))
        self._cb_range = None
        if exchange_range is not None:
            row_bytes, base = 4 * P.shape[1], G.data_ptr()

            def call_range(_ctx, k, ptr, n, stream):
                try:
                    exchange_range(int(k), (int(ptr) - base) // row_bytes, int(n) // P.shape[1], int(stream or 0))
                    return 0
                except Exception as e:                 # noqa: BLE001 -- must not unwind through the C frames
                    self._exc = e
                    return 1
            self._cb_range = EXCHANGE_RANGE_FN(call_range)
        self._exc = None
        cfg = TrainerConfig(
            P=_dev(P, torch.float32, "P"), Q=_dev(Q, torch.float32, "Q"), G=_dev(G, torch.float32, "G"),
            num_users=P.shape[0], num_items=int(num_items) if num_items is not None else Q.shape[0], d=P.shape[1], lr=float(lr),
            indptr=_dev(indptr, torch.int64, "indptr"), indices=_dev(indices, torch.int32, "indices"),
            batch=self.batch, seed=int(seed) & (2**64 - 1), seed_key=int(seed_key) & (2**64 - 1),
            neg_block=int(neg_block), two_pass=int(bool(two_pass)),
            sample_ws=C.c_void_p(self.sample_ws.data_ptr()) if self.sample_ws is not None else None,
            sample_ws_bytes=self.sample_ws.numel() if self.sample_ws is not None else 0,
            user_sig=ptr(user_sig, torch.int64, "user_sig"), item_cdf=ptr(item_cdf, torch.int32, "item_cdf"),
            triplets=_dev(self.triplets, torch.int32, "triplets"),
            hot_slot=ptr(hot.slot if hot else None, torch.int32, "hot slot"),
            G_hot=ptr(hot.ghot if hot else None, torch.float32, "ghot"),
            hot_items=ptr(hot.items if hot else None, torch.int32, "hot items"),
            n_hot=hot.n if hot else 0, hot_replicas=hot.replicas if hot else 0,
            loss_acc=ptr(loss_acc, torch.float32, "loss_acc"),
            exchange_begin=self._cb[0] or EXCHANGE_FN(), exchange_end=self._cb[1] or EXCHANGE_FN(), exchange_ctx=None,
            exchange_applies=int(bool(exchange_applies)), sort_min_batch=int(sort_min_batch), step0=int(step0), epoch_pos0=int(epoch_pos0),
            G_alt=ptr(G_alt, torch.float32, "G_alt"), stale_exchange=int(G_alt is not None),   # opt-in: one step stale
            exchange_kind=int(exchange_kind), comm=comm.handle if comm is not None else None,
            item_rows_padded=int(item_rows_padded), chunks=self.chunks, reserved0=0, items_real=int(items_real),
            chunk_pos=ptr(self.chunk_pos, torch.int64, "chunk_pos"), progress=ptr(self.progress, torch.int32, "progress"),
            exchange_range=self._cb_range or EXCHANGE_RANGE_FN(), mesh=mesh.handle if mesh is not None else None,
            csc=csc.handle if csc is not None else None,
            touched=C.c_void_p(self.touched.data_ptr()) if self.touched is not None else None)
        self._h = C.c_void_p()
        _check(lib().rsx_bpr_trainer_create(C.byref(cfg), C.byref(self._h)), "rsx_bpr_trainer_create")

    def run(self, n_steps, batch=None, global_batch=None, time_every=0):
        batch = self.batch if batch is None else int(batch)
        rc = lib().rsx_bpr_trainer_run(self._h, int(n_steps), batch, int(global_batch or batch), int(time_every), _stream())
        if self._exc is not None:
            e, self._exc = self._exc, None
            raise e
        _check(rc, "rsx_bpr_trainer_run")

    def state(self):
        step, pos = C.c_int64(), C.c_int64()
        _check(lib().rsx_bpr_trainer_state(self._h, C.byref(step), C.byref(pos)), "rsx_bpr_trainer_state")
        return step.value, pos.value

    def seek(self, step, epoch_pos):
        _check(lib().rsx_bpr_trainer_seek(self._h, int(step), int(epoch_pos), _stream()), "rsx_bpr_trainer_seek")

    def last_batch(self):
        """(u, i, j) views of the triplets the most recent step consumed, its neg_block and neg_key"""
        pu, pi, pj = C.c_void_p(), C.c_void_p(), C.c_void_p()
        b, nb, key = C.c_int64(), C.c_int(), C.c_uint64()
        _check(lib().rsx_bpr_trainer_last_batch(self._h, C.byref(pu), C.byref(pi), C.byref(pj), C.byref(b), C.byref(nb),
                                                C.byref(key)), "rsx_bpr_trainer_last_batch")
        base, es = self.triplets.data_ptr(), 4
        view = lambda p: self.triplets[(p.value - base) // es:(p.value - base) // es + b.value]
        return view(pu), view(pi), view(pj), nb.value, key.value

    def check(self):
        """after a chunked run: raises if a triplet left its item range or a wait on the step kernel timed out (synchronises)"""
        _check(lib().rsx_bpr_trainer_check(self._h, _stream()), "rsx_bpr_trainer_check")

    def last_chunk_pos(self):
        """(chunked) first batch position of every item range in the batch the most recent step consumed"""
        pu = C.c_void_p()
        _check(lib().rsx_bpr_trainer_last_batch(self._h, C.byref(pu), None, None, None, None, None), "rsx_bpr_trainer_last_batch")
        slot = (pu.value - self.triplets.data_ptr()) // 4 // (3 * self.batch)
        return self.chunk_pos[slot * (self.chunks + 1):(slot + 1) * (self.chunks + 1)]

    def kernel_ms(self):
        """mean duration of the timed step kernels of the last run (waits for them)"""
        ms, n = C.c_double(), C.c_int64()
        _check(lib().rsx_bpr_trainer_kernel_ms(self._h, C.byref(ms), C.byref(n)), "rsx_bpr_trainer_kernel_ms")
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rsx_bpr_trainer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                              # noqa: BLE001 -- interpreter shutdown
            pass


def _mask_ptrs(mask):
    if mask is None:
        return None, None
    indptr, indices = mask
    return _dev(indptr, torch.int64, "mask indptr"), _dev(indices, torch.int32, "mask indices")


def score(P, Q, user_ids, mask=None, out=None):
    """S = P[user_ids] @ Q.T on the matrix cores, -inf at mask CSR positions."""
    rows, I = user_ids.numel(), Q.shape[0]
    if out is None:
        out = torch.empty((rows, I), dtype=torch.float32, device=P.device)
    mp, mi = _mask_ptrs(mask)
    _check(lib().rsx_score(_dev(P, torch.float32, "P"), _dev(user_ids, torch.int32, "user_ids"), rows,
                           _dev(Q, torch.float32, "Q"), I, P.shape[1], mp, mi,
                           _dev(out, torch.float32, "out"), _stream()), "rsx_score")
    return out


def topk(scores, K, want_values=False):
    rows, I = scores.shape
    idx = torch.empty((rows, K), dtype=torch.int32, device=scores.device)
    val = torch.empty((rows, K), dtype=torch.float32, device=scores.device) if want_values else None
    _check(lib().rsx_topk(_dev(scores, torch.float32, "scores"), rows, I, K, _dev(idx, torch.int32, "idx"),
                          _dev(val, torch.float32, "val") if val is not None else None, _stream()), "rsx_topk")
    return (idx, val) if want_values else idx


def score_topk(P, Q, user_ids, K, mask=None, want_values=False, ws=None):
    rows, I = user_ids.numel(), Q.shape[0]
    need = lib().rsx_score_topk_workspace_d(rows, I, Q.shape[1])
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=P.device)
    idx = torch.empty((rows, K), dtype=torch.int32, device=P.device)
    val = torch.empty((rows, K), dtype=torch.float32, device=P.device) if want_values else None
    mp, mi = _mask_ptrs(mask)
    _check(lib().rsx_score_topk(
        _dev(P, torch.float32, "P"), _dev(user_ids, torch.int32, "user_ids"), rows,
        _dev(Q, torch.float32, "Q"), I, P.shape[1], mp, mi, K, _dev(idx, torch.int32, "idx"),
        _dev(val, torch.float32, "val") if val is not None else None,
        C.c_void_p(ws.data_ptr()), ws.numel() * ws.element_size(), _stream()), "rsx_score_topk")
    return (idx, val) if want_values else idx
