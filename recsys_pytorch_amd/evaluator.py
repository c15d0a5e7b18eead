"""Host-side mirror of evaluation/evaluator.py:10-54 (holdout and leave_one_out protocols).

    Evaluator(eval_input, eval_target, protocol, ks).evaluate(model) -> {'Prec@5': ...}

Flow of the reference: model.eval(); predict -> float32 -> per-row top-max_k
(evaluation/backend/cython/include/func.h:12-31) -> per-user Prec/Recall/NDCG
(holdout.h:20-103) -> float32 mean over users (utils/stats.py:30-32).
Here the [U x I] score matrix never leaves the device: `model.predict_topk`
(MFMA scoring + mask + device top-K) returns only [U x max_k] indices, and the
metric loop is the C++ host function rsx_eval_holdout behind the C ABI.
Top-K runs over ALL U rows, including users without targets, as the reference
does (evaluator.py:31-37); such users are skipped in the mean
(the reference would divide by zero there).
protocol 'leave_one_out' (evaluation/backend/__init__.py:18-21 routes it to loo.h:19-85 / loo.py:11-32): HR@K and
NDCG@K of ONE held-out item per user -- the FIRST target of the user's row (loo.py:20 `target[u][0]`); the metric
loop is rsx_eval_loo.  A user without a target makes the reference raise IndexError there; such users are skipped.
"""
from collections.abc import Iterable

import numpy as np
import scipy.sparse as sp

HOLDOUT_METRICS = ['Prec', 'Recall', 'NDCG']   # evaluation/backend/__init__.py:1
LOO_METRICS = ['HR', 'NDCG']                   # evaluation/backend/__init__.py:2


class Evaluator:
    def __init__(self, eval_input, eval_target, protocol, ks, eval_batch_size=1024):
        self.top_k = sorted(list(ks)) if isinstance(ks, Iterable) else [ks]
        self.max_k = max(self.top_k)
        self.batch_size = eval_batch_size
        self.eval_input = eval_input
        self.eval_target = sp.csr_matrix(eval_target)
        self.eval_target.sort_indices()
        if protocol not in ('holdout', 'leave_one_out'):          # evaluation/backend/__init__.py:18-21
            raise KeyError(protocol)
        self.protocol = protocol

    def evaluate(self, model, mean=True):
        from . import rsx
        model.eval()
        num_users = self.eval_target.shape[0]
        eval_users = np.arange(num_users)                      # sparse_to_dict keys (utils/types.py:13-21)
        if hasattr(model, "predict_topk"):
            # (the package's models hand the ids back in a pinned buffer they keep -- consumed right here, before their next call)
            kw = {"reuse_host": True} if getattr(model, "topk_reuse_host", False) else {}
            pred = model.predict_topk(eval_users, self.eval_input, self.max_k, self.batch_size, **kw)
        else:   # any reference-style model: dense predict, top-k on the device
            import torch
            output = model.predict(eval_users, self.eval_input, self.batch_size)
            pred = rsx.topk(torch.from_numpy(output.astype(np.float32)).cuda(), self.max_k).cpu().numpy()
        has_target = np.diff(self.eval_target.indptr) > 0
        if self.protocol == 'leave_one_out':
            first = np.minimum(self.eval_target.indptr[:-1], max(len(self.eval_target.indices) - 1, 0))
            truth = np.where(has_target, self.eval_target.indices[first] if len(self.eval_target.indices) else -1, -1)
            res = rsx.eval_loo(pred, self.top_k, truth)
            metrics = LOO_METRICS
        else:
            res = rsx.eval_holdout(pred, self.top_k, self.eval_target.indptr, self.eval_target.indices)
            metrics = HOLDOUT_METRICS
        scores = {}
        for m, metric in enumerate(metrics):
            for q, k in enumerate(self.top_k):
                col = res[has_target, m * len(self.top_k) + q]
                scores['%s@%d' % (metric, k)] = np.mean(col, dtype=np.float32) if mean else col.tolist()
        return scores
