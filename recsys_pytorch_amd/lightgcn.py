"""Host-side mirror of models/LightGCN.py for the accelerated path.

    LightGCN(dataset, hparams, device); fit(dataset, exp_config, evaluator, early_stop, loggers);
    predict(eval_users, eval_pos, test_batch_size); registry name `LightGCN` (models/__init__.py:15)

What the reference does per batch (models/LightGCN.py:83-87,117-123): propagate the WHOLE graph
(L sparse products, :188-197), mean over the L+1 layers, BPR loss on the propagated tables,
autograd back through the L products, Adam(lr 1e-3) on the base tables (:46).  Here:
  propagate  rsx_spmm_csr x L (+ running layer sum) + rsx_scale          csrc/rsx_graph.hip
  loss/grad  rsx_bpr_grad on the propagated tables (dense dOut)          csrc/rsx_bpr.hip
  backward   A_hat is symmetric: dE0 = mean_k A_hat^k dOut  ->  rsx_spmm_csr x L again
  optimizer  rsx_adam_apply on E0 = [P; Q]                               (reference-exact Adam)
hparams keys as conf/LightGCN.yaml: emb_dim, num_layers, node_dropout (must be 0: the reference's
dropout path crashes), split (ignored: one CSR), reg (read but unused by the reference,
:33).  The normalised adjacency is built on the host with scipy like the reference (:228-258).
Triplets come from the device sampler (see mf.py); `train_step` replays explicit triplets.
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from .data import csr_to_device
from .mf import BaseModel, IdsToHost, _get, _pad_dim, device_mask, end_of_epoch, kept_for_matrix


def normalized_adjacency(train_csr):
    """A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 (models/LightGCN.py:241-252), CSR float32"""
    R = sp.csr_matrix(train_csr, dtype=np.float32)
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5)
    dinv[np.isinf(dinv)] = 0.0
    D = sp.diags(dinv.astype(np.float32))
    A = (D @ A @ D).tocsr().astype(np.float32)
    A.sort_indices()
    return A


class LightGCN(BaseModel):
    def __init__(self, dataset, hparams, device):
        super().__init__()
        from . import rsx
        self._k = rsx
        self.num_users, self.num_items = dataset.num_users, dataset.num_items
        self.emb_dim = int(hparams["emb_dim"])
        self.num_layers = int(hparams["num_layers"])
        if float(_get(hparams, "node_dropout", 0.0)) > 0:
            raise NotImplementedError("node_dropout > 0 crashes in the reference too (LightGCN.py:165,182)")
        self.lr = float(_get(hparams, "lr", 1e-3))                      # LightGCN.py:46
        self.seed = int(_get(hparams, "seed", 2020))
        self.device = torch.device(device)
        d, dp = self.emb_dim, _pad_dim(self.emb_dim)
        self._dpad = dp
        N = self.num_users + self.num_items
        E0 = torch.zeros(N, dp, dtype=torch.float32)
        E0[:, :d].normal_(0, 0.01)                                      # LightGCN.py:50-51
        self._E0 = E0.to(self.device).contiguous()
        self._m, self._v = torch.zeros_like(self._E0), torch.zeros_like(self._E0)
        self._out = torch.zeros_like(self._E0)                          # propagated tables (mean of layers)
        self._dout = torch.zeros_like(self._E0)                         # dL/dOut, dense
        self._g = torch.zeros_like(self._E0)
        self._ta, self._tb = torch.zeros_like(self._E0), torch.zeros_like(self._E0)
        U = self.num_users
        self.user_embedding = nn.Embedding(U, d, _weight=self._E0[:U, :d])
        self.item_embedding = nn.Embedding(self.num_items, d, _weight=self._E0[U:, :d])
        self.user_embedding.weight.requires_grad_(False)
        self.item_embedding.weight.requires_grad_(False)
        self.Graph = None
        self._t = 0
        self._fresh = False            # is self._out the propagation of the current E0?
        self._nz = None                # per-row "non-zero" flags of dL/dOut (first backward product)

    # -- tables / graph ------------------------------------------------------------------------
    def load_tables(self, P, Q):
        d, U = self.emb_dim, self.num_users
        self._E0.zero_()
        self._E0[:U, :d] = torch.as_tensor(np.asarray(P), dtype=torch.float32).to(self.device)
        self._E0[U:, :d] = torch.as_tensor(np.asarray(Q), dtype=torch.float32).to(self.device)
        self._fresh = False

    def getSparseGraph(self, rating_matrix, adjacency=None):
        """models/LightGCN.py:228-266; `adjacency` lets a caller hand in a prebuilt A_hat"""
        A = normalized_adjacency(rating_matrix) if adjacency is None else adjacency
        # (d: the row width of the products -- lets the longest rows of a popularity-skewed graph go by scatter, rsx_spmm_hot_rows)
        self.Graph = self._k.SpmmGraph(A, self.device, d=self._dpad) if hasattr(self._k, "lib") else self._k.SpmmGraph(A, self.device)
        return self.Graph

    def _idx(self, t):
        return torch.as_tensor(t).to(device=self.device, dtype=torch.int32).contiguous()

    # -- models/LightGCN.py:174-202 ----------------------------------------------------------------
    def _propagate(self, src, acc, src_nonzero=None, out_wanted=None, scale=True):
        """acc = mean_{k=0..L} A_hat^k src   (src untouched).  src_nonzero (uint8 per row, 0 = the row of src is entirely
        zero) lets the FIRST product skip the fetch of such rows (bit-identical; include/rsx.h: rsx_spmm_csr_sparse_rows);
        out_wanted (uint8 per row) lets the LAST product compute only the rows of acc the caller will read (the others are then
        NOT the propagation: include/rsx.h: rsx_spmm_csr_select_rows); scale = False leaves out the 1 / (L + 1) of the layer mean
        (a caller who has folded it into src: the product is linear)"""
        k = self._k
        if self.Graph is None:
            raise RuntimeError("no graph yet: call fit() or getSparseGraph(train_matrix) first "
                               "(the reference builds it in fit, models/LightGCN.py:70)")
        native = hasattr(k, "lib")
        fuse_first = native and (self.num_layers > 1 or out_wanted is None)      # the first product starts the running sum: acc = src + A src
        if not fuse_first:
            acc.copy_(src)
        cur, nxt = src, self._ta
        for layer in range(self.num_layers):
            if layer == 0 and fuse_first:
                k.spmm(self.Graph, cur, nxt, S_acc=acc, x_nonzero=src_nonzero, S_init=src)
            elif layer == 0 and src_nonzero is not None and native:
                k.spmm(self.Graph, cur, nxt, S_acc=acc, x_nonzero=src_nonzero)
            elif layer == self.num_layers - 1 and out_wanted is not None and native:
                k.spmm(self.Graph, cur, nxt, S_acc=acc, y_wanted=out_wanted)
            else:
                k.spmm(self.Graph, cur, nxt, S_acc=acc)
            cur, nxt = nxt, (self._tb if nxt is self._ta else self._ta)
        if scale:
            k.scale(acc, 1.0 / (self.num_layers + 1))

    def update_lightgcn_embedding(self):
        self._propagate(self._E0, self._out)
        U = self.num_users
        self.user_embeddings, self.item_embeddings = self._out[:U], self._out[U:]
        self._fresh = True

    def forward(self, user_ids, item_ids):
        if not self._fresh:
            self.update_lightgcn_embedding()
        return self._k.pair_score(self._out[:self.num_users], self._out[self.num_users:], self._idx(user_ids),
                                  self._idx(item_ids))

    # -- one training step: LightGCN.py:83-87 (zero_grad, process_one_batch, backward, Adam step) --
    def train_step(self, users, pos, neg):
        k, U = self._k, self.num_users
        u, i, j = self._idx(users), self._idx(pos), self._idx(neg)
        # the rows of the stacked [users; items] table this batch touches: the loss reads the propagated tables there and nowhere
        # else (models/LightGCN.py:117-123), and dL/dOut is non-zero there and nowhere else
        if self._nz is None:
            self._nz = torch.zeros(self._E0.shape[0], dtype=torch.uint8, device=self.device)
        k.mark_batch_rows(self._nz, u, i, j, U)
        # forward: L products, the last one only for the batch's rows (65 536 of 1M users in a batch: 93 % of its user rows unread)
        self._propagate(self._E0, self._out, out_wanted=self._nz, scale=False)
        k.scale_rows(self._out, self._nz, 1.0 / (self.num_layers + 1))      # the layer mean, where the loss reads it
        self._fresh = False               # (self._out holds the propagation at the batch's rows only)
        acc = torch.zeros(k.RSX_LOSS_SLOTS, dtype=torch.float32, device=self.device)
        # dL/dOut carries the 1 / (L + 1) of the layer mean already (the propagation is linear: mean_k A^k (a g) = a mean_k A^k g), so
        # the backward pass needs no sweep over the table to scale its result
        alpha = 1.0 / (self.num_layers + 1)
        k.bpr_grad(self._out[:U], self._out[U:], self._dout[:U], self._dout[U:], u, i, j, alpha / max(1, u.numel()),
                   loss_acc=acc)
        # back through the L products; the first one is told which rows of dL/dOut to fetch at all
        self._propagate(self._dout, self._g, src_nonzero=self._nz, scale=False)
        k.scale_rows(self._dout, self._nz, 0.0)     # dL/dOut is non-zero in the batch's rows only: clear those, not the table
        self._t += 1
        k.adam_apply(self._E0, self._m, self._v, self._g, self.lr, self._t)
        self._fresh = False
        return acc.sum() / max(1, u.numel())

    def process_one_batch(self, users, pos_items, neg_items):
        """loss only (LightGCN.py:117-123)"""
        k, U = self._k, self.num_users
        self.update_lightgcn_embedding()
        acc = torch.zeros(k.RSX_LOSS_SLOTS, dtype=torch.float32, device=self.device)
        u = self._idx(users)
        k.bpr_step(self._out[:U], self._out[U:], None, u, self._idx(pos_items), self._idx(neg_items), 0.0, 1.0,
                   loss_acc=acc, no_update=True)
        return acc.sum() / max(1, u.numel())

    # -- models/LightGCN.py:68-115 ---------------------------------------------------------------------
    def fit(self, dataset, exp_config, evaluator=None, early_stop=None, loggers=None):
        from .sharded import BPREngine
        train_matrix = dataset.train_data
        # (the graph -- 1.9 s of host work at 1M x 100K -- and the device copy of the matrix are kept while the caller hands in the same matrix)
        self.Graph = kept_for_matrix(self, "_graph_cache", train_matrix, lambda: self.getSparseGraph(train_matrix))
        indptr, indices = device_mask(self, train_matrix, "_train_cache")
        batch_size = int(_get(exp_config, "batch_size"))
        num_epochs = int(_get(exp_config, "num_epochs"))
        verbose = _get(exp_config, "verbose", 0)
        test_from, test_step = int(_get(exp_config, "test_from", 1)), int(_get(exp_config, "test_step", 1))
        sampler = BPREngine(self._E0[:self.num_users], self._E0[self.num_users:], self.lr, seed=self.seed)
        n_data = self.num_users
        num_batches = int(np.ceil(n_data / batch_size))
        scores = None
        for epoch in range(1, num_epochs + 1):
            self.train()
            epoch_loss = torch.zeros((), dtype=torch.float32, device=self.device)
            sampler.epoch_pos = (epoch - 1) * n_data
            for b in range(num_batches):
                bsz = min(batch_size, n_data - b * batch_size)
                u, i, j = sampler.sample(indptr, indices, bsz)
                sampler.step_count += 1
                batch_loss = self.train_step(u, i, j)          # (stays on the device: read back for the progress line and the loggers only)
                epoch_loss += batch_loss
                if verbose and b % 50 == 0:
                    print('(%3d / %3d) loss = %.4f' % (b, num_batches, float(batch_loss)))
            scores, stop = end_of_epoch(self, epoch, {'loss': float(epoch_loss)} if loggers is not None else {}, scores, evaluator, early_stop,
                                        loggers, test_from, test_step)
            if stop:
                break
        return {'scores': early_stop.best_score if early_stop is not None else scores}

    # -- models/LightGCN.py:125-150 -------------------------------------------------------------------------
    def predict_batch_users(self, user_ids):
        if not self._fresh:
            self.update_lightgcn_embedding()
        return self._k.score(self._out[:self.num_users], self._out[self.num_users:], self._idx(user_ids))

    def predict(self, eval_users, eval_pos, test_batch_size):
        self.update_lightgcn_embedding()                                # LightGCN.py:131
        eval_users = np.asarray(eval_users)
        pred_matrix = np.zeros(eval_pos.shape)
        mask = csr_to_device(eval_pos, self.device)
        U = self.num_users
        for s in range(0, len(eval_users), test_batch_size):
            batch_users = eval_users[s:s + test_batch_size]
            S = self._k.score(self._out[:U], self._out[U:], self._idx(batch_users), mask=mask)
            pred_matrix[batch_users] = S.cpu().numpy()
        return pred_matrix

    topk_reuse_host = True          # predict_topk takes reuse_host (evaluator.py)

    def predict_topk(self, eval_users, eval_pos, K, test_batch_size=1024, want_values=False, reuse_host=False):
        self.update_lightgcn_embedding()
        eval_users = np.asarray(eval_users)
        mask = device_mask(self, eval_pos)
        U = self.num_users
        out = []
        to_host = IdsToHost(self, len(eval_users), K) if (reuse_host and self.device.type == "cuda" and len(eval_users)) else None
        # large catalogs take the fused path, which wants many 8 192-row passes per call (like MF.predict_topk)
        chunk = max(int(test_batch_size), 65536) if self.num_items >= 32768 else int(test_batch_size)
        ws = None
        users_dev = self._idx(eval_users)           # once (mf.py: predict_topk)
        for s in range(0, len(eval_users), chunk):
            users = users_dev[s:s + chunk]
            if ws is None and hasattr(self._k, "lib"):
                need = self._k.lib().rsx_score_topk_workspace_d(users.numel(), self.num_items, self._dpad)
                ws = torch.empty(max(need, 4) // 4 + 64, dtype=torch.float32, device=self.device)
            r = self._k.score_topk(self._out[:U], self._out[U:], users, K, mask=mask, want_values=False, ws=ws)
            if to_host is not None:
                to_host.put(s, r)
            else:
                out.append(r.cpu().numpy())
        if to_host is not None:
            return to_host.done()
        return np.concatenate(out) if out else np.zeros((0, K), np.int32)
