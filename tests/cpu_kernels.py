"""TEST INFRASTRUCTURE: a CPU stand-in for the `kernels` argument of
recsys_pytorch_amd.sharded.BPREngine, backed by the oracle (oracle/mf_oracle.c).
It exists so the sharding / collective logic can run under gloo without a GPU.
It is never importable from the product package."""
import ctypes as C

import numpy as np
import torch

import oracle

RSX_LOSS_SLOTS = 2048


def bpr_step_workspace(num_users, max_batch, d):
    return 16


def bpr_step(P, Q, G, u, i, j, lr, inv_batch, loss_acc=None, users_unique=False, ws=None, no_update=False,
             hot=None, neg_block=0, neg_key=0, only=None, wide_offsets=False, deterministic=False, batch_sorted=False):
    """same contract as include/rsx.h:rsx_bpr_step, on host tensors: G += dQ (scaled by
    inv_batch), P -= lr*dP, loss slots += sum softplus(-x)."""
    Pn, Qn = P.numpy(), Q.numpy()
    B = int(u.numel())
    keep = (i.numpy() >= 0)
    uu, ii, jj = (t.numpy().astype(np.int64)[keep] for t in (u, i, j))
    gP, gQ = np.zeros_like(Pn), np.zeros_like(Qn)
    loss = C.c_double(0)
    if len(uu):
        oracle.lib().orc_bpr_grad(Pn, Qn, uu, ii, jj, len(uu), Pn.shape[1], gP, gQ, C.byref(loss))
    scale = float(inv_batch) * len(uu)      # orc_bpr_grad uses 1/len(batch); rescale to inv_batch
    if loss_acc is not None and only != "users":
        loss_acc[0] += float(loss.value) * len(uu)
    if no_update:
        return
    if only != "users":
        G += torch.from_numpy(gQ * np.float32(scale))
    if only != "items":
        P -= torch.from_numpy(gP * np.float32(scale * lr))


def apply_item_grad(Q, G, lr, hot=None):
    Q -= lr * G
    G.zero_()


def bpr_sample(indptr, indices, num_items, batch, seed, step, epoch_pos, u_out, i_out, j_out, neg_block=0,
               neg_key=0, sort_pos=False, ws=None, user_sig=None, item_cdf=None):
    """simple host sampler with the same guarantees (unique users, true pos, true neg)"""
    U = indptr.numel() - 1
    rng = np.random.default_rng(seed + 1000003 * step)
    perm = np.random.default_rng(seed + (epoch_pos // U)).permutation(U)
    ip, ix = indptr.numpy(), indices.numpy()
    for b in range(batch):
        u = perm[(epoch_pos + b) % U]
        row = ix[ip[u]:ip[u + 1]]
        u_out[b] = int(u)
        if len(row) == 0:
            i_out[b] = -1; j_out[b] = -1
            continue
        i_out[b] = int(row[rng.integers(len(row))])
        while True:
            j = int(rng.integers(num_items))
            if j not in row:
                break
        j_out[b] = j


def pointwise_grad(P, Q, GP, GQ, u, i, y, inv_n, loss_func="ce", loss_acc=None):
    """same contract as include/rsx.h:rsx_pointwise_grad on host tensors (oracle-backed)"""
    Pn, Qn = P.numpy(), Q.numpy()
    gP, gQ = np.zeros_like(Pn), np.zeros_like(Qn)
    loss = C.c_double(0)
    n = int(u.numel())
    oracle.lib().orc_pointwise_grad(Pn, Qn, u.numpy().astype(np.int64), i.numpy().astype(np.int64),
                                    np.ascontiguousarray(y.numpy(), np.float32), n, Pn.shape[1], int(loss_func == "mse"),
                                    gP, gQ, C.byref(loss))
    scale = np.float32(float(inv_n) * n)        # the oracle carries 1/n; rescale to inv_n
    GP += torch.from_numpy(gP * scale)
    GQ += torch.from_numpy(gQ * scale)
    if loss_acc is not None:
        loss_acc[0] += float(loss.value) * n


def adam_apply(W, M, V, G, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam's single-tensor update (the formulas of oracle/mf_oracle.c:orc_adam_apply), then G = 0"""
    bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
    M += (1.0 - beta1) * (G - M)
    V.mul_(beta2).addcmul_(G, G, value=1.0 - beta2)
    W -= (lr / bc1) * (M / (V.sqrt() / bc2 ** 0.5 + eps))
    G.zero_()


# -- the relabelled item space of the item-range pipelines (BPREngine.set_chunks): only its HOST logic runs on the CPU ----------
def chunk_rows(items_real, chunks, neg_block):
    """include/rsx.h:rsx_chunk_rows"""
    most = -(-items_real // chunks)
    return -(-most // neg_block) * neg_block


def build_signature(indptr, indices, neg_block):
    return None


def build_item_cdf(indptr, indices, num_items):
    return None


class HotItems:
    def __init__(self, item_counts, num_hot, replicas, d, device):
        counts = torch.as_tensor(item_counts)
        self.items = torch.topk(counts, int(min(num_hot, counts.numel()))).indices.to(torch.int32)
        self.n, self.replicas = int(self.items.numel()), int(replicas)
