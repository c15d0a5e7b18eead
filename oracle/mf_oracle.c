/*
 * oracle/mf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's BPR-MF hot path
 * (yoongi0428/RecSys_PyTorch).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker.
 * The shipped path (recsys_pytorch_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED -- every function here is checked against golden
 * vectors produced by importing the reference itself in the build container
 * (oracle/gen_golden.py -> tests/golden/ *.npz; tests/test_oracle_golden.py).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ---- forward: models/MF.py:32-42 ---------------------------------------
 * embeddings(): gather P[u], Q[i]; forward(): sum(mul(user_emb,item_emb),1).
 * torch's fp32 CPU reduction is a vectorised tree; a plain left-to-right fp32
 * sum differs in the last ulps only, far inside the 1e-5 contract.           */
static float orc_dot(const float *a, const float *b, int d)
{
    float s = 0.0f;
    for (int k = 0; k < d; ++k) s += a[k] * b[k];
    return s;
}

/* ---- BPR loss of one batch: models/MF.py:99-107 --------------------------
 * x_b = r(u,i) - r(u,j);  loss = -mean_b log(sigmoid(x_b))                   */
ORC_EXPORT double orc_bpr_loss(const float *P, const float *Q,
                               const int64_t *u, const int64_t *i, const int64_t *j,
                               int64_t B, int d)
{
    double acc = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        const float *pu = P + u[b] * d;
        float x = orc_dot(pu, Q + i[b] * d, d) - orc_dot(pu, Q + j[b] * d, d);
        float s = 1.0f / (1.0f + expf(-x));          /* F.sigmoid, MF.py:105 */
        acc += -(double)logf(s);                      /* .log()              */
    }
    return B > 0 ? acc / (double)B : 0.0;             /* .mean()             */
}

/* ---- dense gradients of that loss: models/MF.py:67 (loss.backward()) ------
 * nn.Embedding(sparse=False) => dense [U x d] / [I x d] grads; duplicates are
 * summed (index_add); every term carries 1/B from the mean (MF.py:105).
 *   dL/dx_b = -(1 - sigmoid(x_b)) / B = -sigmoid(-x_b)/B
 *   dP[u_b] += dL/dx_b * (Q[i_b] - Q[j_b])
 *   dQ[i_b] += dL/dx_b * P[u_b];   dQ[j_b] -= dL/dx_b * P[u_b]
 * gP / gQ must be zero-filled by the caller (optimizer.zero_grad, MF.py:64). */
ORC_EXPORT void orc_bpr_grad(const float *P, const float *Q,
                             const int64_t *u, const int64_t *i, const int64_t *j,
                             int64_t B, int d, float *gP, float *gQ, double *loss_out)
{
    double acc = 0.0;
    const float invB = 1.0f / (float)B;
    for (int64_t b = 0; b < B; ++b) {
        const float *pu = P + u[b] * d;
        const float *qi = Q + i[b] * d;
        const float *qj = Q + j[b] * d;
        float x = orc_dot(pu, qi, d) - orc_dot(pu, qj, d);
        float s = 1.0f / (1.0f + expf(-x));
        acc += -(double)logf(s);
        float g = -(1.0f - s) * invB;                 /* dL/dx_b             */
        float *gpu = gP + u[b] * d;
        float *gqi = gQ + i[b] * d;
        float *gqj = gQ + j[b] * d;
        for (int k = 0; k < d; ++k) {
            /* autograd accumulates the pos-forward and neg-forward
             * contributions separately (two embedding backward calls)       */
            gpu[k] += g * qi[k];
            gpu[k] += -g * qj[k];
            gqi[k] += g * pu[k];
            gqj[k] += -g * pu[k];
        }
    }
    if (loss_out) *loss_out = B > 0 ? acc / (double)B : 0.0;
}

/* ---- one training step with SGD: models/MF.py:64-68 with the optimizer
 * attribute swapped to torch.optim.SGD(lr) (north-star optimizer; the
 * reference's loss/backward path is untouched).  theta -= lr * grad, applied
 * to EVERY row (dense), all gradients taken at the pre-step tables.
 * scratch gP [U*d], gQ [I*d] supplied by the caller.                         */
ORC_EXPORT void orc_bpr_step_sgd(float *P, float *Q, int64_t U, int64_t I,
                                 const int64_t *u, const int64_t *i, const int64_t *j,
                                 int64_t B, int d, float lr,
                                 float *gP, float *gQ, double *loss_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(gQ, 0, sizeof(float) * (size_t)I * d);
    orc_bpr_grad(P, Q, u, i, j, B, d, gP, gQ, loss_out);
    for (int64_t n = 0; n < U * d; ++n) P[n] -= lr * gP[n];
    for (int64_t n = 0; n < I * d; ++n) Q[n] -= lr * gQ[n];
}

/* ---- one training step with the optimizer as shipped: models/MF.py:30
 * torch.optim.Adam(lr=1e-3, betas=(0.9,0.999), eps=1e-8, weight_decay=0),
 * dense over ALL rows: a row whose moments are non-zero keeps moving even
 * when its gradient is zero this step.  t = step count AFTER increment.
 * Follows torch's single-tensor Adam:
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
 *   step = lr / (1 - b1^t);  denom = sqrt(v)/sqrt(1 - b2^t) + eps
 *   theta -= step * m / denom                                               */
static void orc_adam_apply(float *w, float *m, float *v, const float *g, int64_t n,
                           double lr, double b1, double b2, double eps, int64_t t)
{
    /* the hyper-parameters are Python floats (doubles) in torch; its kernels receive each scalar rounded ONCE to the tensor's
     * type: lerp_'s weight 1 - b1, mul_'s b2, addcmul_'s value 1 - b2 = 0.001 (not 1.0f - 0.999f = 0.00099998713: round 5) */
    double bc1 = 1.0 - pow(b1, (double)t);
    double bc2 = 1.0 - pow(b2, (double)t);
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - b1), b2f = (float)b2, w2 = (float)(1.0 - b2), epsf = (float)eps;
    for (int64_t k = 0; k < n; ++k) {
        m[k] = m[k] + w1 * (g[k] - m[k]);                      /* lerp_         */
        v[k] = b2f * v[k] + w2 * g[k] * g[k];                  /* addcmul_      */
        float denom = sqrtf(v[k]) / bc2_sqrt + epsf;
        w[k] = w[k] - step_size * (m[k] / denom);              /* addcdiv_      */
    }
}

ORC_EXPORT void orc_bpr_step_adam(float *P, float *Q, int64_t U, int64_t I,
                                  const int64_t *u, const int64_t *i, const int64_t *j,
                                  int64_t B, int d, double lr, double b1, double b2, double eps,
                                  int64_t t,
                                  float *mP, float *vP, float *mQ, float *vQ,
                                  float *gP, float *gQ, double *loss_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(gQ, 0, sizeof(float) * (size_t)I * d);
    orc_bpr_grad(P, Q, u, i, j, B, d, gP, gQ, loss_out);
    orc_adam_apply(P, mP, vP, gP, U * d, lr, b1, b2, eps, t);
    orc_adam_apply(Q, mQ, vQ, gQ, I * d, lr, b1, b2, eps, t);
}

/* ---- the POINTWISE branch: models/MF.py:99-102 with hparams['pointwise'] = True -----------------
 * loss = loss_func(forward(users, items), ratings), loss_func = F.mse_loss when hparams['loss_func'] == 'mse',
 * else F.binary_cross_entropy_with_logits (MF.py:21); both with reduction 'mean'.  x_b = <P[u_b], Q[i_b]>:
 *   ce : l_b = max(x,0) - x y + log(1 + exp(-|x|))   (torch's stable form)   dl/dx = sigmoid(x) - y
 *   mse: l_b = (x - y)^2                                                       dl/dx = 2 (x - y)
 * dense grads like the pairwise branch (nn.Embedding(sparse=False)): dP[u_b] += dl/dx / n * Q[i_b],
 * dQ[i_b] += dl/dx / n * P[u_b]; users AND items repeat inside a batch (data/generators.py:105-130 puts
 * batch_size interactions plus one sampled negative for EVERY user into each batch).  loss_kind 0 = ce, 1 = mse. */
ORC_EXPORT void orc_pointwise_grad(const float *P, const float *Q, const int64_t *u, const int64_t *i,
                                   const float *y, int64_t n, int d, int loss_kind, float *gP, float *gQ,
                                   double *loss_out)
{
    double acc = 0.0;
    const float invn = 1.0f / (float)n;
    for (int64_t b = 0; b < n; ++b) {
        const float *pu = P + u[b] * d;
        const float *qi = Q + i[b] * d;
        const float x = orc_dot(pu, qi, d);
        float g;
        if (loss_kind == 1) {
            acc += (double)((x - y[b]) * (x - y[b]));
            g = 2.0f * (x - y[b]) * invn;
        } else {
            acc += (double)(fmaxf(x, 0.0f) - x * y[b] + log1pf(expf(-fabsf(x))));
            g = (1.0f / (1.0f + expf(-x)) - y[b]) * invn;
        }
        float *gpu = gP + u[b] * d;
        float *gqi = gQ + i[b] * d;
        for (int k = 0; k < d; ++k) {
            gpu[k] += g * qi[k];
            gqi[k] += g * pu[k];
        }
    }
    if (loss_out) *loss_out = n > 0 ? acc / (double)n : 0.0;
}

ORC_EXPORT void orc_pointwise_step_sgd(float *P, float *Q, int64_t U, int64_t I, const int64_t *u, const int64_t *i,
                                       const float *y, int64_t n, int d, int loss_kind, float lr,
                                       float *gP, float *gQ, double *loss_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(gQ, 0, sizeof(float) * (size_t)I * d);
    orc_pointwise_grad(P, Q, u, i, y, n, d, loss_kind, gP, gQ, loss_out);
    for (int64_t k = 0; k < U * d; ++k) P[k] -= lr * gP[k];
    for (int64_t k = 0; k < I * d; ++k) Q[k] -= lr * gQ[k];
}

ORC_EXPORT void orc_pointwise_step_adam(float *P, float *Q, int64_t U, int64_t I, const int64_t *u, const int64_t *i,
                                        const float *y, int64_t n, int d, int loss_kind, double lr, double b1, double b2,
                                        double eps, int64_t t, float *mP, float *vP, float *mQ, float *vQ,
                                        float *gP, float *gQ, double *loss_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(gQ, 0, sizeof(float) * (size_t)I * d);
    orc_pointwise_grad(P, Q, u, i, y, n, d, loss_kind, gP, gQ, loss_out);
    orc_adam_apply(P, mP, vP, gP, U * d, lr, b1, b2, eps, t);
    orc_adam_apply(Q, mQ, vQ, gQ, I * d, lr, b1, b2, eps, t);
}

/* ---- full-catalog scoring: models/MF.py:109-112 --------------------------
 * S[r, :] = P[users[r]] @ Q.T  (fp32), out row-major [Bu x I].               */
ORC_EXPORT void orc_score(const float *P, const int64_t *users, int64_t Bu,
                          const float *Q, int64_t I, int d, float *out)
{
    for (int64_t r = 0; r < Bu; ++r) {
        const float *pu = P + users[r] * d;
        float *o = out + r * I;
        for (int64_t c = 0; c < I; ++c) o[c] = orc_dot(pu, Q + c * d, d);
    }
}

/* ---- seen-item masking: models/MF.py:130 ---------------------------------
 * pred_matrix[eval_pos.nonzero()] = -inf ; rows are indexed by USER ID.
 * Here: scores is [Bu x I] for users[r]; CSR (indptr int64, indices int32).  */
ORC_EXPORT void orc_mask_seen(float *scores, const int64_t *users, int64_t Bu, int64_t I,
                              const int64_t *indptr, const int32_t *indices)
{
    for (int64_t r = 0; r < Bu; ++r) {
        int64_t uu = users[r];
        for (int64_t p = indptr[uu]; p < indptr[uu + 1]; ++p)
            scores[r * I + indices[p]] = -INFINITY;
    }
}

/* ---- per-row top-k: evaluation/backend/cython/include/func.h:12-31 -------
 * std::partial_sort_copy of an iota index vector with comparator
 * ratings[x1] > ratings[x2]: the K largest scores, sorted descending.  The
 * reference leaves the order among EXACTLY tied scores unspecified (heap
 * order in C++, introselect+argsort in the numpy twin python/func.py:4-17);
 * this restatement breaks ties by the lower item index, which is one of the
 * admissible answers.  out: int32 [rows x K].                                */
typedef struct { float s; int32_t i; } orc_pair;

static int orc_better(orc_pair a, orc_pair b)   /* a ranks before b */
{
    if (a.s > b.s) return 1;
    if (a.s < b.s) return 0;
    return a.i < b.i;
}

static void orc_sift_down(orc_pair *h, int n, int k)
{   /* min-heap on "rank": root = worst of the kept K */
    for (;;) {
        int l = 2 * k + 1, r = l + 1, w = k;
        if (l < n && orc_better(h[w], h[l])) w = l;
        if (r < n && orc_better(h[w], h[r])) w = r;
        if (w == k) return;
        orc_pair t = h[k]; h[k] = h[w]; h[w] = t; k = w;
    }
}

static int orc_cmp_desc(const void *a, const void *b)
{
    orc_pair x = *(const orc_pair *)a, y = *(const orc_pair *)b;
    return orc_better(x, y) ? -1 : (orc_better(y, x) ? 1 : 0);
}

ORC_EXPORT void orc_topk(const float *scores, int64_t I, int64_t rows, int K, int32_t *out)
{
    orc_pair *h = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)K);
    for (int64_t r = 0; r < rows; ++r) {
        const float *s = scores + r * I;
        int n = 0;
        for (int64_t c = 0; c < I; ++c) {
            orc_pair e = { s[c], (int32_t)c };
            if (n < K) {
                h[n++] = e;
                if (n == K) for (int k = K / 2 - 1; k >= 0; --k) orc_sift_down(h, K, k);
            } else if (orc_better(e, h[0])) {
                h[0] = e; orc_sift_down(h, K, 0);
            }
        }
        qsort(h, (size_t)n, sizeof(orc_pair), orc_cmp_desc);
        for (int k = 0; k < n; ++k) out[r * K + k] = h[k].i;
        for (int k = n; k < K; ++k) out[r * K + k] = -1;
    }
    free(h);
}

/* ---- holdout metrics: evaluation/backend/cython/include/holdout.h:20-103 -
 * Prec@K = hits/K, Recall@K = hits/truth_len, NDCG@K = DCG/iDCG with
 * 1/log2(i+2); float accumulators as in the header (hits, DCG, iDCG float).
 * results layout [user][metric*K_len + k], metrics = Prec, Recall, NDCG.
 * truth given as CSR (indptr int64, indices int32) instead of int**.         */
ORC_EXPORT void orc_holdout(int64_t users_num, const int32_t *rankings, int max_k,
                            const int32_t *Ks, int K_len,
                            const int64_t *t_indptr, const int32_t *t_indices,
                            float *results)
{
    for (int64_t uid = 0; uid < users_num; ++uid) {
        const int32_t *rk = rankings + uid * max_k;
        const int32_t *truth = t_indices + t_indptr[uid];
        int truth_len = (int)(t_indptr[uid + 1] - t_indptr[uid]);
        float *res = results + uid * 3 * K_len;
        float hits = 0, iDCG = 0, DCG = 0;
        for (int p = 0; p < max_k; ++p) {
            int found = 0;
            for (int t = 0; t < truth_len; ++t) if (truth[t] == rk[p]) { found = 1; break; }
            if (found) { hits += 1; DCG += 1.0 / log2(p + 2); }
            if (p < truth_len) iDCG += 1.0 / log2(p + 2);
            for (int q = 0; q < K_len; ++q) {
                if (Ks[q] == p + 1) {
                    res[0 * K_len + q] = hits / (float)Ks[q];
                    res[1 * K_len + q] = hits / truth_len;
                    res[2 * K_len + q] = DCG / iDCG;
                }
            }
        }
    }
}

/* ---- leave-one-out metrics: evaluation/backend/cython/include/loo.h:19-85 (python twin
 * evaluation/backend/python/loo.py:11-32).  One held-out item per user (truth_len = 1: only the FIRST target counts,
 * loo.h:31, loo.py:20): hit_at = 1-based position of that item in the user's ranking, max_k + 1 when absent;
 * HR@K = [K >= hit_at], NDCG@K = 1 / log2(hit_at + 1) when K >= hit_at else 0.
 * results layout [user][metric*K_len + k], metrics = HR, NDCG (loo.h:64-84).                                   */
ORC_EXPORT void orc_loo(int64_t users_num, const int32_t *rankings, int max_k, const int32_t *Ks, int K_len,
                        const int32_t *truth, float *results)
{
    for (int64_t uid = 0; uid < users_num; ++uid) {
        const int32_t *rk = rankings + uid * max_k;
        int hit_at = max_k + 1;
        for (int p = 0; p < max_k; ++p) if (rk[p] == truth[uid]) { hit_at = p + 1; break; }
        float *res = results + uid * 2 * K_len;
        for (int q = 0; q < K_len; ++q) {
            if (Ks[q] >= hit_at) { res[q] = 1.0f; res[K_len + q] = (float)(1 / log2(hit_at + 1)); }
            else { res[q] = 0.0f; res[K_len + q] = 0.0f; }
        }
    }
}

/* ==== LightGCN (BASELINE config 5) ==================================
 * models/LightGCN.py:174-202 (_lightgcn_embedding): all_emb = cat(user_w, item_w);
 * L times all_emb = A_hat @ all_emb (torch.sparse.mm, :196); output = mean over the L+1 layers
 * (:198-200).  A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 is built on the host by the reference
 * (:228-258) and handed over here as CSR.                                                    */
ORC_EXPORT void orc_spmm_csr(const int64_t *indptr, const int32_t *indices, const float *vals,
                             const float *X, float *Y, int64_t N, int d)
{
    for (int64_t r = 0; r < N; ++r) {
        float *y = Y + r * d;
        for (int k = 0; k < d; ++k) y[k] = 0.0f;
        for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            const float a = vals[p];
            const float *x = X + (int64_t)indices[p] * d;
            for (int k = 0; k < d; ++k) y[k] += a * x[k];
        }
    }
}

/* out = mean_{k=0..L} A_hat^k E0 ; tmpA/tmpB scratch [N*d] */
ORC_EXPORT void orc_lightgcn_propagate(const int64_t *indptr, const int32_t *indices, const float *vals,
                                       const float *E0, float *out, float *tmpA, float *tmpB,
                                       int64_t N, int d, int L)
{
    const float *cur = E0;
    float *nxt = tmpA;
    for (int64_t n = 0; n < N * d; ++n) out[n] = E0[n];
    for (int l = 0; l < L; ++l) {
        orc_spmm_csr(indptr, indices, vals, cur, nxt, N, d);
        for (int64_t n = 0; n < N * d; ++n) out[n] += nxt[n];
        cur = nxt;
        nxt = (nxt == tmpA) ? tmpB : tmpA;
    }
    const float inv = 1.0f / (float)(L + 1);      /* torch.mean over the stacked layers */
    for (int64_t n = 0; n < N * d; ++n) out[n] *= inv;
}

/* One training step, models/LightGCN.py:83-87,117-123 with the optimizer as shipped (:46, Adam
 * lr 1e-3): propagate, BPR loss on the propagated tables, backward through the L sparse products
 * (A_hat is symmetric: dE0 = mean_k A_hat^k dOut), dense Adam on E0 = [P;Q].
 * scratch: out, dout, tA, tB, g: [N*d] each.                                                  */
ORC_EXPORT void orc_lightgcn_step_adam(float *E0, float *mE, float *vE, int64_t U, int64_t I,
                                       const int64_t *indptr, const int32_t *indices, const float *vals,
                                       int L, const int64_t *u, const int64_t *i, const int64_t *j,
                                       int64_t B, int d, double lr, double b1, double b2, double eps, int64_t t,
                                       float *out, float *dout, float *tA, float *tB, float *g,
                                       double *loss_out)
{
    const int64_t N = U + I;
    orc_lightgcn_propagate(indptr, indices, vals, E0, out, tA, tB, N, d, L);
    memset(dout, 0, sizeof(float) * (size_t)N * d);
    /* dense gradients w.r.t. the PROPAGATED tables: same closed form as MF (orc_bpr_grad) */
    orc_bpr_grad(out, out + U * d, u, i, j, B, d, dout, dout + U * d, loss_out);
    /* back through the propagation */
    const float *cur = dout;
    float *nxt = tA;
    for (int64_t n = 0; n < N * d; ++n) g[n] = dout[n];
    for (int l = 0; l < L; ++l) {
        orc_spmm_csr(indptr, indices, vals, cur, nxt, N, d);
        for (int64_t n = 0; n < N * d; ++n) g[n] += nxt[n];
        cur = nxt;
        nxt = (nxt == tA) ? tB : tA;
    }
    const float inv = 1.0f / (float)(L + 1);
    for (int64_t n = 0; n < N * d; ++n) g[n] *= inv;
    orc_adam_apply(E0, mE, vE, g, N * d, lr, b1, b2, eps, t);
}
