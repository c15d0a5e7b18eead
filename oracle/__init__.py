"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU checker for the BPR-MF hot path: ctypes doorway onto ``liboracle.so``
(plain-C restatement, ``oracle/mf_oracle.c``) and, when it was built,
``_ref/libref_eval.so`` (the reference's own ``func.h`` / ``holdout.h``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``recsys_pytorch_amd`` never does.

Parity status: pinned against golden vectors generated from the imported
reference (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(with_ref=True):
    """Compile the checker (gcc).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if with_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        L = C.CDLL(path)
        L.orc_bpr_loss.restype = C.c_double
        L.orc_bpr_loss.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64p, C.c_int64, C.c_int]
        L.orc_bpr_grad.restype = None
        L.orc_bpr_grad.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64p, C.c_int64, C.c_int,
                                   _f32p, _f32p, C.POINTER(C.c_double)]
        L.orc_bpr_step_sgd.restype = None
        L.orc_bpr_step_sgd.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, _i64p, _i64p, _i64p,
                                       C.c_int64, C.c_int, C.c_float, _f32p, _f32p,
                                       C.POINTER(C.c_double)]
        L.orc_bpr_step_adam.restype = None
        L.orc_bpr_step_adam.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, _i64p, _i64p, _i64p,
                                        C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_int64, _f32p, _f32p, _f32p, _f32p,
                                        _f32p, _f32p, C.POINTER(C.c_double)]
        L.orc_score.restype = None
        L.orc_score.argtypes = [_f32p, _i64p, C.c_int64, _f32p, C.c_int64, C.c_int, _f32p]
        L.orc_mask_seen.restype = None
        L.orc_mask_seen.argtypes = [_f32p, _i64p, C.c_int64, C.c_int64, _i64p, _i32p]
        L.orc_topk.restype = None
        L.orc_topk.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, _i32p]
        L.orc_holdout.restype = None
        L.orc_holdout.argtypes = [C.c_int64, _i32p, C.c_int, _i32p, C.c_int, _i64p, _i32p, _f32p]
        L.orc_loo.restype = None
        L.orc_loo.argtypes = [C.c_int64, _i32p, C.c_int, _i32p, C.c_int, _i32p, _f32p]
        L.orc_spmm_csr.restype = None
        L.orc_spmm_csr.argtypes = [_i64p, _i32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int]
        L.orc_lightgcn_propagate.restype = None
        L.orc_lightgcn_propagate.argtypes = [_i64p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64,
                                             C.c_int, C.c_int]
        L.orc_lightgcn_step_adam.restype = None
        L.orc_lightgcn_step_adam.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_int64, _i64p, _i32p, _f32p,
                                             C.c_int, _i64p, _i64p, _i64p, C.c_int64, C.c_int, C.c_double,
                                             C.c_double, C.c_double, C.c_double, C.c_int64, _f32p, _f32p, _f32p,
                                             _f32p, _f32p, C.POINTER(C.c_double)]
        L.orc_pointwise_grad.restype = None
        L.orc_pointwise_grad.argtypes = [_f32p, _f32p, _i64p, _i64p, _f32p, C.c_int64, C.c_int, C.c_int, _f32p, _f32p,
                                         C.POINTER(C.c_double)]
        L.orc_pointwise_step_sgd.restype = None
        L.orc_pointwise_step_sgd.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, _i64p, _i64p, _f32p, C.c_int64, C.c_int,
                                             C.c_int, C.c_float, _f32p, _f32p, C.POINTER(C.c_double)]
        L.orc_pointwise_step_adam.restype = None
        L.orc_pointwise_step_adam.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, _i64p, _i64p, _f32p, C.c_int64, C.c_int,
                                              C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                              _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own native code (oracle/_ref), or None if it was not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_eval.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_top_k_array_index.restype = None
        R.ref_top_k_array_index.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _i32p]
        R.ref_evaluate_holdout.restype = None
        R.ref_evaluate_holdout.argtypes = [C.c_int, _i32p, C.c_int, _i32p, C.c_int, _i64p, _i32p, _f32p]
        if hasattr(R, "ref_evaluate_loo"):
            R.ref_evaluate_loo.restype = None
            R.ref_evaluate_loo.argtypes = [C.c_int, _i32p, C.c_int, _i32p, C.c_int, _i32p, _f32p]
        _REF = R
    return _REF


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class MFOracle:
    """Stateful CPU model: two fp32 tables + optimizer state (models/MF.py:14-30)."""

    def __init__(self, P0, Q0, optimizer="sgd", lr=0.05, betas=(0.9, 0.999), eps=1e-8):
        self.P = np.array(P0, dtype=np.float32, order="C", copy=True)
        self.Q = np.array(Q0, dtype=np.float32, order="C", copy=True)
        self.U, self.d = self.P.shape
        self.I = self.Q.shape[0]
        self.optimizer, self.lr, self.betas, self.eps = optimizer, float(lr), betas, float(eps)
        self._gP = np.zeros_like(self.P)
        self._gQ = np.zeros_like(self.Q)
        self.t = 0
        if optimizer == "adam":
            self.mP, self.vP = np.zeros_like(self.P), np.zeros_like(self.P)
            self.mQ, self.vQ = np.zeros_like(self.Q), np.zeros_like(self.Q)

    def grad(self, u, i, j):
        gP, gQ = np.zeros_like(self.P), np.zeros_like(self.Q)
        loss = C.c_double(0)
        lib().orc_bpr_grad(self.P, self.Q, _i64(u), _i64(i), _i64(j), len(u), self.d,
                           gP, gQ, C.byref(loss))
        return gP, gQ, loss.value

    def step(self, u, i, j):
        loss = C.c_double(0)
        self.t += 1
        if self.optimizer == "sgd":
            lib().orc_bpr_step_sgd(self.P, self.Q, self.U, self.I, _i64(u), _i64(i), _i64(j),
                                   len(u), self.d, self.lr, self._gP, self._gQ, C.byref(loss))
        else:
            lib().orc_bpr_step_adam(self.P, self.Q, self.U, self.I, _i64(u), _i64(i), _i64(j),
                                    len(u), self.d, self.lr, self.betas[0], self.betas[1],
                                    self.eps, self.t, self.mP, self.vP, self.mQ, self.vQ,
                                    self._gP, self._gQ, C.byref(loss))
        return loss.value

    # the pointwise branch (models/MF.py:99-102): loss_kind "ce" (binary_cross_entropy_with_logits) or "mse"
    def pointwise_grad(self, u, i, y, loss_kind="ce"):
        gP, gQ = np.zeros_like(self.P), np.zeros_like(self.Q)
        loss = C.c_double(0)
        lib().orc_pointwise_grad(self.P, self.Q, _i64(u), _i64(i), np.ascontiguousarray(y, np.float32), len(u), self.d,
                                 int(loss_kind == "mse"), gP, gQ, C.byref(loss))
        return gP, gQ, loss.value

    def pointwise_step(self, u, i, y, loss_kind="ce"):
        loss = C.c_double(0)
        self.t += 1
        y = np.ascontiguousarray(y, np.float32)
        if self.optimizer == "sgd":
            lib().orc_pointwise_step_sgd(self.P, self.Q, self.U, self.I, _i64(u), _i64(i), y, len(u), self.d,
                                         int(loss_kind == "mse"), self.lr, self._gP, self._gQ, C.byref(loss))
        else:
            lib().orc_pointwise_step_adam(self.P, self.Q, self.U, self.I, _i64(u), _i64(i), y, len(u), self.d,
                                          int(loss_kind == "mse"), self.lr, self.betas[0], self.betas[1], self.eps,
                                          self.t, self.mP, self.vP, self.mQ, self.vQ, self._gP, self._gQ, C.byref(loss))
        return loss.value

    def score(self, users):
        users = _i64(users)
        out = np.empty((len(users), self.I), dtype=np.float32)
        lib().orc_score(self.P, users, len(users), self.Q, self.I, self.d, out)
        return out


def score(P, Q, users):
    P = np.ascontiguousarray(P, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    users = _i64(users)
    out = np.empty((len(users), Q.shape[0]), dtype=np.float32)
    lib().orc_score(P, users, len(users), Q, Q.shape[0], P.shape[1], out)
    return out


def mask_seen(scores, users, indptr, indices):
    lib().orc_mask_seen(scores, _i64(users), scores.shape[0], scores.shape[1],
                        _i64(indptr), np.ascontiguousarray(indices, np.int32))
    return scores


def topk(scores, K):
    scores = np.ascontiguousarray(scores, np.float32)
    out = np.empty((scores.shape[0], K), dtype=np.int32)
    lib().orc_topk(scores, scores.shape[1], scores.shape[0], K, out)
    return out


def ref_topk(scores, K):
    scores = np.ascontiguousarray(scores, np.float32)
    out = np.zeros((scores.shape[0], K), dtype=np.int32)
    ref_lib().ref_top_k_array_index(scores, scores.shape[1], scores.shape[0], K, out)
    return out


def holdout(rankings, Ks, t_indptr, t_indices, use_ref=False):
    rankings = np.ascontiguousarray(rankings, np.int32)
    Ks = np.ascontiguousarray(Ks, np.int32)
    n = rankings.shape[0]
    res = np.zeros((n, 3 * len(Ks)), dtype=np.float32)
    ti = np.ascontiguousarray(t_indices, np.int32)
    tp = _i64(t_indptr)
    if use_ref:
        ref_lib().ref_evaluate_holdout(n, rankings, rankings.shape[1], Ks, len(Ks), tp, ti, res)
    else:
        lib().orc_holdout(n, rankings, rankings.shape[1], Ks, len(Ks), tp, ti, res)
    return res


def loo(rankings, Ks, truth, use_ref=False):
    """HR / NDCG of the leave-one-out protocol, [users x 2*len(Ks)] (loo.h:19-85); truth: one item per user"""
    rankings = np.ascontiguousarray(rankings, np.int32)
    Ks = np.ascontiguousarray(Ks, np.int32)
    truth = np.ascontiguousarray(truth, np.int32)
    n = rankings.shape[0]
    res = np.zeros((n, 2 * len(Ks)), dtype=np.float32)
    if use_ref:
        ref_lib().ref_evaluate_loo(n, rankings, rankings.shape[1], Ks, len(Ks), truth, res)
    else:
        lib().orc_loo(n, rankings, rankings.shape[1], Ks, len(Ks), truth, res)
    return res


def normalized_adjacency(train_csr):
    """A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 as CSR float32 (models/LightGCN.py:228-258), built the
    direct way with scipy (the reference goes through lil/dok matrices; same matrix)."""
    import scipy.sparse as sp
    R = sp.csr_matrix(train_csr, dtype=np.float32)
    U, I = R.shape
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5)
    dinv[np.isinf(dinv)] = 0.0
    D = sp.diags(dinv.astype(np.float32))
    A = (D @ A @ D).tocsr().astype(np.float32)
    A.sort_indices()
    return A


class LightGCNOracle:
    """CPU LightGCN: base table E0 = [P;Q], Adam as shipped (models/LightGCN.py:46)."""

    def __init__(self, P0, Q0, A_hat, num_layers, lr=1e-3):
        self.U, self.d = P0.shape
        self.I = Q0.shape[0]
        self.N = self.U + self.I
        self.E0 = np.ascontiguousarray(np.concatenate([P0, Q0]), np.float32)
        self.m, self.v = np.zeros_like(self.E0), np.zeros_like(self.E0)
        self.indptr = _i64(A_hat.indptr)
        self.indices = np.ascontiguousarray(A_hat.indices, np.int32)
        self.vals = np.ascontiguousarray(A_hat.data, np.float32)
        self.L, self.lr, self.t = int(num_layers), float(lr), 0
        self._s = [np.zeros_like(self.E0) for _ in range(5)]

    def propagate(self):
        out = np.empty_like(self.E0)
        lib().orc_lightgcn_propagate(self.indptr, self.indices, self.vals, self.E0, out, self._s[0], self._s[1],
                                     self.N, self.d, self.L)
        return out[:self.U], out[self.U:]

    def step(self, u, i, j):
        self.t += 1
        loss = C.c_double(0)
        lib().orc_lightgcn_step_adam(self.E0, self.m, self.v, self.U, self.I, self.indptr, self.indices,
                                     self.vals, self.L, _i64(u), _i64(i), _i64(j), len(u), self.d, self.lr,
                                     0.9, 0.999, 1e-8, self.t, *self._s, C.byref(loss))
        return loss.value

    @property
    def P(self):
        return self.E0[:self.U]

    @property
    def Q(self):
        return self.E0[self.U:]
