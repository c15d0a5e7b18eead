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
    neg_block = max(1, neg_block)
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


class BPRTrainer:
    """CPU stand-in for recsys_pytorch_amd.rsx.BPRTrainer in its CHUNKED form (include/rsx.h: "item chunks"): the same
    schedule as csrc/rsx_train.hip's chunked branch -- per step: sample with the range rule, then range by range
    [step over the range's positions -> exchange_range(k) -> apply(k)] -- on host tensors, oracle-backed.  It exists so that
    BPREngine's N > 1 chunked path (the relabelled item space agreed between the ranks, the per-range collective on a view of
    G, 1 / sum_r B_r, adopt / sync_items) runs under gloo with two ranks and no GPU."""

    def __init__(self, P, Q, G, indptr, indices, lr, batch, seed, seed_key, neg_block=0, hot=None, user_sig=None, item_cdf=None,
                 loss_acc=None, comm=None, exchange_kind=0, chunks=0, items_real=0, step0=0, epoch_pos0=0, exchange_range=None,
                 **other):
        assert chunks > 1 and comm is None and not other, "the CPU stand-in only runs the chunked loop with exchange_range"
        self.P, self.Q, self.G, self.lr = P, Q, G, float(lr)
        self.ip, self.ix = indptr.numpy(), indices.numpy()
        self.batch, self.chunks, self.items_real = int(batch), int(chunks), int(items_real)
        self.Ic = chunk_rows(items_real, chunks, neg_block)
        assert Q.shape[0] == self.chunks * self.Ic
        base, rem = divmod(self.items_real, self.chunks)
        self.n_real = [base + (k < rem) for k in range(self.chunks)]
        self.seed, self.step, self.epoch_pos = int(seed), int(step0), int(epoch_pos0)
        self.loss_acc, self.exchange_range = loss_acc, exchange_range
        self._last = None

    def _sample(self, B):
        U = len(self.ip) - 1
        if (self.epoch_pos % U) + B > U:
            self.epoch_pos = (self.epoch_pos // U + 1) * U
        rng = np.random.default_rng(self.seed + 1000003 * self.step)
        perm = np.random.default_rng(self.seed + 17 * (self.epoch_pos // U)).permutation(U)
        us = perm[(self.epoch_pos % U):(self.epoch_pos % U) + B]
        u, i, j = [], [], []
        for a in us:
            row = self.ix[self.ip[a]:self.ip[a + 1]]
            if len(row) == 0:
                continue
            pos = int(row[rng.integers(len(row))])
            k = pos // self.Ic
            cand = np.setdiff1d(k * self.Ic + np.arange(self.n_real[k]), row)      # the real items of the positive's range
            if len(cand) == 0:
                continue                                                            # (a user owning its whole range: skipped)
            u.append(int(a)); i.append(pos); j.append(int(cand[rng.integers(len(cand))]))
        order = np.argsort(np.asarray(i, np.int64), kind="stable")
        u, i, j = (np.asarray(x, np.int64)[order] for x in (u, i, j))
        cp = np.searchsorted(i, np.arange(self.chunks + 1) * self.Ic).astype(np.int64)
        self.epoch_pos += B
        return u, i, j, cp

    def run(self, n_steps, batch=None, global_batch=None, time_every=0):
        B = self.batch if batch is None else int(batch)
        inv = 1.0 / float(global_batch or B)
        Pn, Qn = self.P.numpy(), self.Q.numpy()
        for _ in range(int(n_steps)):
            u, i, j, cp = self._sample(B)
            for k in range(self.chunks):
                uu, ii, jj = (np.ascontiguousarray(x[cp[k]:cp[k + 1]]) for x in (u, i, j))
                assert np.all(ii // self.Ic == k) and np.all(jj // self.Ic == k)
                lo, hi = k * self.Ic, (k + 1) * self.Ic
                if len(uu):
                    gP, gQ = np.zeros_like(Pn), np.zeros_like(Qn)
                    loss = C.c_double(0)
                    oracle.lib().orc_bpr_grad(Pn, Qn, uu, ii, jj, len(uu), Pn.shape[1], gP, gQ, C.byref(loss))
                    scale = np.float32(inv * len(uu))                               # orc_bpr_grad carries 1 / len(batch)
                    assert not gQ[:lo].any() and not gQ[hi:].any()                   # range k's triplets touch range k's rows only
                    self.G[lo:hi] += torch.from_numpy(gQ[lo:hi] * scale)
                    self.P -= torch.from_numpy(gP * (scale * np.float32(self.lr)))
                    if self.loss_acc is not None:
                        self.loss_acc[0] += float(loss.value) * len(uu)
                if self.exchange_range is not None:
                    self.exchange_range(k, lo, self.Ic, 0)
                self.Q[lo:hi] -= self.lr * self.G[lo:hi]
                self.G[lo:hi].zero_()
            self._last = (u, i, j, cp)
            self.step += 1

    def state(self):
        return self.step, self.epoch_pos

    def last_batch(self):
        u, i, j, _ = self._last
        return torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j), 0, 0

    def last_chunk_pos(self):
        return torch.from_numpy(self._last[3])

    def check(self):
        pass

    def close(self):
        pass
