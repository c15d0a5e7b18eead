"""Host-side mirror of the reference's model interface for the BPR-MF path.

    models/BaseModel.py:3-14   BaseModel(nn.Module): forward / fit / predict
    models/MF.py:13-132        MF(dataset, hparams, device), fit(...), predict(...)
    models/__init__.py:12      registry name `MF`  (main.py:46-47 getattr(models, name))

Same constructor convention, same `fit(dataset, exp_config, evaluator, early_stop,
loggers) -> {'scores': ...}` loop shape, same `predict(eval_users, eval_pos,
test_batch_size) -> ndarray[U x I]` with -inf at seen items.  The arithmetic is NOT
torch: every step/score goes through librsx.so (include/rsx.h).  torch tensors are
storage.  There is no CPU fallback.

Divergences from the reference, all documented in DESIGN.md:
  * optimizer: hparams['optimizer'] = 'sgd' (default, north star; lr default 0.05) or 'adam'
    (the reference's dense Adam, MF.py:30, reference-exact; lr default 1e-3).
  * triplets: sampled on the device every step (include/rsx.h:rsx_bpr_sample), true
    BPR sampling, instead of PairwiseGenerator's once-per-fit host sampling with its
    quirks (data/generators.py:165,182-185).  `train_step` replays explicit triplets.
  * hparams['neg_block'] (default 8; 0 = off): when a batch holds >= 2 triplets per item the sampler
    orders it by positive item and stratifies the negatives by item block so that the step kernel can
    sum item gradients on chip; tests/test_gpu_model.py checks it trains as well.
    hparams['neg_block_min'] (default 2): the smallest block the engine may pick.  Same expectation as independent negatives, more
    variance per step the smaller the block: measured on a planted-factor dataset (profiles/r06_sampler_quality.txt) NDCG@10 falls short
    of independent negatives' by 2.4 % at block 2 and 0.8 % at block 8 with a LARGE step size (0.1 per triplet), by nothing measurable
    at the step size where that model is best (0.02); larger blocks cost a few percent of speed (sharded.py: pick_neg_block).
  * hparams['pointwise'] = True (MF.py:48-51,101-102): the pointwise branch, hparams['loss_func'] 'mse' or anything
    else = binary cross entropy with logits (MF.py:21).  Batches are the reference generator's (data/generators.py:
    105-130: batch_size interactions of a per-epoch permutation PLUS one uniformly drawn negative, rating 0, for EVERY
    user of the matrix -- its sample_negatives ignores the batch it is handed), assembled on the device; the gradient
    of a batch is rsx_pointwise_grad, the update the as-shipped Adam or the SGD sweep.
  * hparams['chunks'] (default 0 = off): > 1 runs the SGD step as a pipeline over that many item ranges when the blocked
    layout engages (include/rsx.h: "item chunks"; BPREngine.set_chunks): the apply -- and, user-sharded, the exchange --
    of a range travels under the rest of the step kernel; negatives come from the range of the sampled positive.  Which items
    share a range is redrawn every hparams['redraw_ranges_every'] steps (default 64; 0 = one relabelling for the whole fit),
    between epochs: over a fit every item meets every other as a negative.
  * hparams['hot_items'] (default 256; 0 = off; SGD): the gradients of that many most popular items of the train matrix go to
    private replica rows (include/rsx.h: hot_slot_dev) and are folded in the apply -- on a popularity-skewed catalog the atomic
    unit otherwise serialises on a few rows (B = 1M on the Zipf bench graph: 625 vs 340 us per step).  Same sums, another order.
  * hidden_dim is padded to 32/64/128 columns of zeros internally (they stay zero).
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from .data import csr_to_device
from .sharded import BPREngine


class BaseModel(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, *input):
        pass

    def fit(self, *input):
        pass

    def predict(self, eval_users, eval_pos, test_batch_size):
        pass


def end_of_epoch(model, epoch, epoch_summary, scores, evaluator, early_stop, loggers, test_from, test_step):
    """what the reference's fit loops do after every epoch (models/MF.py:75-95, LightGCN.py:93-113):
    evaluate every `test_step` epochs from `test_from`, log the summary, ask early_stop.
    Returns (latest scores, should_stop)."""
    evaluated = evaluator is not None and epoch >= test_from and epoch % test_step == 0
    if evaluated:
        scores = evaluator.evaluate(model)
        epoch_summary.update(scores)
    if loggers is not None:
        for logger in loggers:
            logger.log_metrics(epoch_summary, epoch=epoch)
    stop = False
    if evaluated and early_stop is not None:
        _, stop = early_stop.step(scores, epoch)
    return scores, stop


def device_mask(model, eval_pos, slot="_mask_cache"):
    """the seen-items matrix of an evaluation on the device.  An Evaluator hands in the SAME matrix at every evaluation
    (evaluation/evaluator.py:16-17 keeps eval_input): the device copy of the last one is kept on the model -- by identity of the object,
    its shape and its number of entries -- instead of 40 ms of slicing, sorting and uploading 20 M entries per evaluation at a million users.
    slot = "_train_cache": the same for the train matrix of `fit` (a caller's own loop of one-epoch fits hands in the same matrix every
    time; the SAME device tensors also let the engine keep its sampler tables, which it files under their identity)"""
    if eval_pos is None:
        return None
    return kept_for_matrix(model, slot, eval_pos, lambda: csr_to_device(eval_pos, model.device))


def kept_for_matrix(model, slot, mat, make):
    """make() once per matrix: the result is kept on the model under `slot` while the caller hands in the SAME scipy object -- identity,
    shape, number of entries and a strided sample of its arrays are compared (an edit in place that the sample sees makes it again)"""
    c = getattr(model, slot, None)
    key = (tuple(mat.shape), int(getattr(mat, "nnz", -1)))
    if hasattr(mat, "indices") and hasattr(mat, "indptr"):
        ix, ip = mat.indices, mat.indptr
        key += (int(ix[::max(1, len(ix) // 4096)].astype(np.int64).sum()), int(ip[::max(1, len(ip) // 4096)].astype(np.int64).sum()))
    if c is not None and c[0] is mat and c[1] == key:
        return c[2]
    made = make()
    setattr(model, slot, (mat, key, made))
    return made


class IdsToHost:
    """the [n x K] int32 results of successive scoring passes -> ONE pinned host buffer kept on the model, each pass's copy on a side stream
    under the NEXT pass's kernels (round 6: sixteen blocking copies and a 200 MB concatenate were 55 ms of a 310 ms evaluation of a million
    users).  The array `done()` returns is a view of that buffer: valid until the model's next such call -- which is why only the package's
    own Evaluator, which consumes it at once, asks for it (predict_topk(reuse_host=True))"""

    def __init__(self, model, n, K):
        buf = getattr(model, "_ids_host", None)
        if buf is None or buf.shape[0] < n or buf.shape[1] != K:
            model._ids_host = buf = torch.empty((max(int(n), 1), int(K)), dtype=torch.int32, pin_memory=True)
        if getattr(model, "_ids_stream", None) is None:
            model._ids_stream = torch.cuda.Stream(device=model.device)
        self.buf, self.side, self.keep, self.n = buf, model._ids_stream, [], int(n)

    def put(self, s, r):
        ev = torch.cuda.Event()
        ev.record()                                  # r is complete once the current stream has come this far
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self.buf[s:s + r.shape[0]].copy_(r, non_blocking=True)
        self.keep.append(r)                          # (alive until the copy has run)

    def done(self):
        self.side.synchronize()
        del self.keep[:]
        return self.buf[:self.n].numpy()


def _pad_dim(d):
    """the kernels are instantiated for rows of 32, 64, 128 and 256 floats; any other hidden_dim (the reference takes any,
    models/MF.py:19,23-24; conf/MF.yaml ships 50) is stored with zero columns behind it, which provably stay zero"""
    for p in (32, 64, 128, 256):
        if d <= p:
            return p
    raise ValueError(f"hidden_dim {d} > 256 is not supported by the HIP kernels")


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError, IndexError):
        return getattr(cfg, key, default)


class MF(BaseModel):
    _said_default_optimizer = False

    def __init__(self, dataset, hparams, device, kernels=None):
        super().__init__()
        self.num_users = dataset.num_users
        self.num_items = dataset.num_items
        self.hidden_dim = int(hparams["hidden_dim"])
        self.pointwise = bool(hparams["pointwise"])
        self.loss_func = "mse" if _get(hparams, "loss_func", "ce") == "mse" else "ce"          # models/MF.py:21
        opt = _get(hparams, "optimizer", None)
        if opt is None:
            # conf/MF.yaml has no such key and the reference trains dense Adam, lr 1e-3 (models/MF.py:30): a maintainer who swaps the
            # class and nothing else gets the north star's SGD -- said once, loudly, not silently
            opt = "sgd"
            if not MF._said_default_optimizer:
                MF._said_default_optimizer = True
                print(f"recsys_pytorch_amd.MF: hparams has no 'optimizer' key -> SGD, lr {float(_get(hparams, 'lr', 0.05))} (the MI355X step kernel; "
                      "the reference's own models/MF.py:30 is dense Adam, lr 1e-3: pass hparams['optimizer'] = 'adam' for that)", flush=True)
        if opt not in ("sgd", "adam"):
            raise ValueError("optimizer must be 'sgd' (north star) or 'adam' (as shipped, models/MF.py:30)")
        self.optimizer_name = opt
        self.lr = float(_get(hparams, "lr", 0.05 if opt == "sgd" else 1e-3))
        self.seed = int(_get(hparams, "seed", 2020))
        # item block of the stratified negatives; 0 = independent uniform negatives always
        self.neg_block = int(_get(hparams, "neg_block", 8))
        self.neg_block_min = int(_get(hparams, "neg_block_min", 2))
        self.chunks = int(_get(hparams, "chunks", 0))
        self.hot_items = int(_get(hparams, "hot_items", 256))
        self.device = torch.device(device)
        self._dpad = _pad_dim(self.hidden_dim)
        d = self.hidden_dim
        # storage: padded row-major fp32 tables; N(0,1) init like nn.Embedding (MF.py:23-24)
        P = torch.zeros(self.num_users, self._dpad, dtype=torch.float32)
        Q = torch.zeros(self.num_items, self._dpad, dtype=torch.float32)
        P[:, :d].normal_()
        Q[:, :d].normal_()
        self._P = P.to(self.device).contiguous()
        self._Q = Q.to(self.device).contiguous()
        self.user_embedding = nn.Embedding(self.num_users, d, _weight=self._P[:, :d])
        self.item_embedding = nn.Embedding(self.num_items, d, _weight=self._Q[:, :d])
        self.user_embedding.weight.requires_grad_(False)
        self.item_embedding.weight.requires_grad_(False)
        self._kernels = kernels
        self._engine = BPREngine(self._P, self._Q, self.lr, kernels=kernels, seed=self.seed, optimizer=opt)
        self._engine.redraw_ranges_every = int(_get(hparams, "redraw_ranges_every", self._engine.redraw_ranges_every))
        self._k = self._engine.k

    # -- tables ---------------------------------------------------------------------
    def load_tables(self, P, Q):
        """overwrite the embedding tables (numpy or tensor [U x d], [I x d])"""
        d = self.hidden_dim
        self._P.zero_(); self._Q.zero_()
        self._P[:, :d] = torch.as_tensor(np.asarray(P), dtype=torch.float32).to(self.device)
        self._Q[:, :d] = torch.as_tensor(np.asarray(Q), dtype=torch.float32).to(self.device)

    def _idx(self, t):
        return torch.as_tensor(t).to(device=self.device, dtype=torch.int32).contiguous()

    # -- models/MF.py:32-42 ------------------------------------------------------------
    def embeddings(self, user_ids, item_ids):
        return self.user_embedding(torch.as_tensor(user_ids).long()), self.item_embedding(torch.as_tensor(item_ids).long())

    def forward(self, user_ids, item_ids):
        return self._k.pair_score(self._P, self._Q, self._idx(user_ids), self._idx(item_ids))

    # -- models/MF.py:99-107: the loss of one batch (no update) ---------------------------
    def process_one_batch(self, users, items, ratings):
        if self.pointwise:                       # MF.py:101-102: loss_func(forward(users, items), ratings)
            u, i = self._idx(users), self._idx(items)
            y = torch.as_tensor(ratings).to(device=self.device, dtype=torch.float32).contiguous()
            acc = torch.zeros(self._k.RSX_LOSS_SLOTS, dtype=torch.float32, device=self.device)
            self._k.pointwise_grad(self._P, self._Q, None, None, u, i, y, 1.0, loss_func=self.loss_func, loss_acc=acc)
            return acc.sum() / max(1, u.numel())
        u, i, j = self._idx(users), self._idx(items), self._idx(ratings)
        acc = torch.zeros(self._k.RSX_LOSS_SLOTS, dtype=torch.float32, device=self.device)
        self._k.bpr_step(self._P, self._Q, None, u, i, j, 0.0, 1.0, loss_acc=acc, no_update=True)
        return acc.sum() / max(1, u.numel())

    # -- models/MF.py:64-68: zero_grad + loss + backward + optimizer.step, fused ------------
    def train_step(self, users, pos, neg, users_unique=False):
        if self.pointwise:                       # (users, items, ratings): users and items may repeat
            u, i = self._idx(users), self._idx(pos)
            y = torch.as_tensor(neg).to(device=self.device, dtype=torch.float32).contiguous()
            acc = self._engine.pointwise_step(u, i, y, self.loss_func)
            return acc.sum() / max(1, u.numel())
        u, i, j = self._idx(users), self._idx(pos), self._idx(neg)
        acc = self._engine.step(u, i, j, users_unique=users_unique)
        return acc.sum() / max(1, u.numel())

    # -- models/MF.py:44-97 ----------------------------------------------------------------
    def fit(self, dataset, exp_config, evaluator=None, early_stop=None, loggers=None):
        train_matrix = dataset.train_data
        indptr, indices = device_mask(self, train_matrix, "_train_cache")
        batch_size = int(_get(exp_config, "batch_size"))
        num_epochs = int(_get(exp_config, "num_epochs"))
        verbose = _get(exp_config, "verbose", 0)
        test_from = int(_get(exp_config, "test_from", 1))
        test_step = int(_get(exp_config, "test_step", 1))
        if self.pointwise:
            return self._fit_pointwise(train_matrix, indptr, indices, batch_size, num_epochs, verbose, test_from, test_step,
                                       evaluator, early_stop, loggers)
        # one triplet per user per epoch, like PairwiseGenerator(num_positives_per_user=1)
        # (data/generators.py:182-195); the last batch of an epoch is short, not dropped (:213)
        n_data = self.num_users
        num_batches = int(np.ceil(n_data / batch_size))
        if self.optimizer_name == "sgd":     # on-chip gradient summation when batch >= 2 * items
            nb = self._engine.set_neg_block(batch_size if self.neg_block > 0 else 0, max(self.neg_block, 1), self.neg_block_min)
            if self.hot_items > 0 and hasattr(self._k, "HotItems"):
                hot_for = getattr(self, "_hot_for", None)           # (the same matrix, the same kernel family: the replica tables of the last fit stand)
                if not (hot_for is not None and hot_for[0] is indices and hot_for[1] == bool(nb) and self._engine.hot is not None):
                    self._engine.set_hot_items(torch.bincount(indices.long(), minlength=self.num_items),
                                               min(self.hot_items, self.num_items))
                    self._hot_for = (indices, bool(nb))
            self._engine.set_chunks(self.chunks if nb else 0)
        scores = None
        # SGD on the HIP library: the batch loop itself is native (include/rsx.h: rsx_bpr_trainer_run),
        # Python is re-entered once per epoch (or every 50 batches when verbose, for the progress line)
        native = self.optimizer_name == "sgd" and hasattr(self._k, "BPRTrainer")
        acc = torch.zeros(self._k.RSX_LOSS_SLOTS, dtype=torch.float32, device=self.device)
        trainer = self._engine.native_trainer(indptr, indices, batch_size, loss_acc=acc) if native else None
        for epoch in range(1, num_epochs + 1):
            self.train()
            epoch_loss = torch.zeros((), dtype=torch.float32, device=self.device)
            self._engine.epoch_pos = (epoch - 1) * n_data
            if native:
                # (a seek drops the batches the trainer sampled ahead -- two of its three slots -- and makes the next step wait for its
                #  sampler: only when the loop really is somewhere else.  After a whole epoch it is not: same step, same position, and the
                #  sampler is a function of those, so the batches sampled ahead ARE the next epoch's.  One-step epochs -- batch = users, the
                #  headline shape -- ran at 0.9 ms per step with the unconditional seek and the per-epoch read of the loss, 3x the step.)
                if trainer.state() != (self._engine.step_count, self._engine.epoch_pos):
                    trainer.seek(self._engine.step_count, self._engine.epoch_pos)
                b = 0
                while b < num_batches:
                    bsz = min(batch_size, n_data - b * batch_size)
                    full_left = (n_data - b * batch_size) // batch_size
                    n = 1 if bsz < batch_size else (min(50, full_left) if verbose else full_left)
                    acc.zero_()
                    trainer.run(n, bsz)
                    chunk_loss = acc.sum() / bsz                    # sum over the n batches of their mean losses
                    epoch_loss += chunk_loss
                    if verbose:
                        print('(%3d / %3d) loss = %.4f' % (b, num_batches, float(chunk_loss) / n))
                    b += n
                self._engine.adopt(trainer)
                if self._engine.relabel_due():      # item ranges: the next epochs pair positives with another 1 / C of the catalog
                    trainer.close()
                    trainer = self._engine.native_trainer(indptr, indices, batch_size, loss_acc=acc)
            for b in range(num_batches if not native else 0):
                bsz = min(batch_size, n_data - b * batch_size)
                step_acc = self._engine.sampled_step(indptr, indices, bsz)
                batch_loss = step_acc.sum() / bsz
                epoch_loss += batch_loss
                if verbose and b % 50 == 0:
                    print('(%3d / %3d) loss = %.4f' % (b, num_batches, float(batch_loss)))
            # (the epoch's loss is read back -- a synchronisation -- only for someone who looks at it: the loggers)
            scores, stop = end_of_epoch(self, epoch, {'loss': float(epoch_loss)} if loggers is not None else {}, scores, evaluator, early_stop,
                                        loggers, test_from, test_step)
            if stop:
                break
        return {'scores': early_stop.best_score if early_stop is not None else scores}

    def _fit_pointwise(self, train_matrix, indptr, indices, batch_size, num_epochs, verbose, test_from, test_step,
                       evaluator, early_stop, loggers):
        """models/MF.py:48-51 + the batches of data/generators.py:58-130 (PointwiseGenerator(return_rating=True,
        num_negatives=1, shuffle=True)): all interactions once per epoch in a fresh permutation, batch_size at a time, and
        EVERY batch extended by one uniformly drawn negative (rating 0) per user of the matrix."""
        dev = self.device
        counts = (indptr[1:] - indptr[:-1])
        users_all = torch.repeat_interleave(torch.arange(self.num_users, device=dev, dtype=torch.int32), counts)
        items_all = indices.to(torch.int32)
        csr = sp.csr_matrix(train_matrix, copy=True)
        csr.sort_indices()                       # the order csr_to_device stores the row in
        ratings_all = torch.as_tensor(np.asarray(csr.data, dtype=np.float32)).to(dev)
        n_data = int(items_all.numel())
        num_batches = int(np.ceil(n_data / batch_size))
        gen = torch.Generator(device=dev)
        gen.manual_seed(self.seed)
        eng = self._engine
        zeros = torch.zeros(self.num_users, dtype=torch.float32, device=dev)
        scores = None
        for epoch in range(1, num_epochs + 1):
            self.train()
            epoch_loss = torch.zeros((), dtype=torch.float32, device=dev)
            perm = torch.randperm(n_data, device=dev, generator=gen)                  # generators.py:107
            for b in range(num_batches):
                idx = perm[b * batch_size:(b + 1) * batch_size]
                # one negative for every user: the pairwise sampler over a batch of ALL users draws, per user, a negative
                # uniform over the items outside the user's row -- the distribution of generators.py:87-91
                eng.epoch_pos = 0
                nu, _, nj = eng.sample(indptr, indices, self.num_users)
                eng.step_count += 1              # a fresh draw next batch
                # a user with an empty (or full) train row has no pairwise sample (j = -1): the reference draws a uniform
                # negative over ALL items for an empty row (generators.py:87-91: prob is all ones) and trains on it
                dead = nj < 0
                if bool(dead.any()):
                    nj = torch.where(dead, torch.randint(0, self.num_items, nj.shape, device=dev, generator=gen,
                                                         dtype=torch.int64).to(torch.int32), nj)
                u = torch.cat([users_all[idx], nu])
                i = torch.cat([items_all[idx], nj])
                y = torch.cat([ratings_all[idx], zeros])
                acc = eng.pointwise_step(u, i, y, self.loss_func, count_step=False)
                batch_loss = acc.sum() / u.numel()
                epoch_loss += batch_loss
                if verbose and b % 50 == 0:
                    print('(%3d / %3d) loss = %.4f' % (b, num_batches, float(batch_loss)))
            scores, stop = end_of_epoch(self, epoch, {'loss': float(epoch_loss)}, scores, evaluator, early_stop,
                                        loggers, test_from, test_step)
            if stop:
                break
        return {'scores': early_stop.best_score if early_stop is not None else scores}

    # -- models/MF.py:109-112 ----------------------------------------------------------------
    def predict_batch_users(self, user_ids):
        return self._k.score(self._P, self._Q, self._idx(user_ids))

    # -- models/MF.py:114-132: dense [U x I] float64 with -inf at eval_pos (small problems) ------
    def predict(self, eval_users, eval_pos, test_batch_size):
        eval_users = np.asarray(eval_users)
        n_bytes = eval_pos.shape[0] * eval_pos.shape[1] * 8
        if n_bytes > 8 << 30:
            raise MemoryError(f"predict() would materialise {n_bytes / 2**30:.0f} GiB on the host "
                              "(models/MF.py:117); use predict_topk() for catalogs this large")
        pred_matrix = np.zeros(eval_pos.shape)
        mask = csr_to_device(eval_pos, self.device)
        for s in range(0, len(eval_users), test_batch_size):
            batch_users = eval_users[s:s + test_batch_size]
            S = self._k.score(self._P, self._Q, self._idx(batch_users), mask=mask)
            pred_matrix[batch_users] = S.cpu().numpy()       # rows indexed by USER ID (MF.py:128)
        return pred_matrix

    # -- the large-catalog twin: only [n x K] indices leave the device -----------------------------
    topk_reuse_host = True          # predict_topk takes reuse_host (evaluator.py)

    def predict_topk(self, eval_users, eval_pos, K, test_batch_size=1024, want_values=False, reuse_host=False):
        """reuse_host: the ids come back as a view of a pinned buffer the model keeps (valid until its next such call) -- see IdsToHost"""
        eval_users = np.asarray(eval_users)
        mask = device_mask(self, eval_pos)
        out_i, out_v = [], []
        to_host = IdsToHost(self, len(eval_users), K) if (reuse_host and not want_values and self.device.type == "cuda" and len(eval_users)) else None
        # large catalogs take the fused path, which wants many 8 192-row passes per call (two are in
        # flight at a time); small ones score a dense [test_batch_size x I] tile like the reference
        chunk = max(int(test_batch_size), 65536) if self.num_items >= 32768 else int(test_batch_size)
        ws = None
        # (the user ids go to the device ONCE: a pageable host-to-device copy per pass -- 512 KB out of the middle of a numpy array --
        #  stalled every third scoring call of a million-user evaluation by 45 ms, tools/eval_time.py)
        users_dev = self._idx(eval_users)
        for s in range(0, len(eval_users), chunk):
            users = users_dev[s:s + chunk]
            if ws is None:
                need = self._k.lib().rsx_score_topk_workspace_d(users.numel(), self.num_items, self._dpad) if hasattr(self._k, "lib") else 0
                ws = torch.empty(max(need, 4) // 4 + 64, dtype=torch.float32, device=self.device)
            r = self._k.score_topk(self._P, self._Q, users, K, mask=mask, want_values=want_values, ws=ws)
            if want_values:
                out_i.append(r[0].cpu().numpy()); out_v.append(r[1].cpu().numpy())
            elif to_host is not None:
                to_host.put(s, r)
            else:
                out_i.append(r.cpu().numpy())
        if to_host is not None:
            return to_host.done()
        idx = np.concatenate(out_i) if out_i else np.zeros((0, K), np.int32)
        return (idx, np.concatenate(out_v)) if want_values else idx
