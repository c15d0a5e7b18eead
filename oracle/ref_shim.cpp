// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" doorway onto the reference's OWN native evaluation code,
// compiled from the headers where they lie under /root/reference (nothing is
// copied into this repository):
//   evaluation/backend/cython/include/func.h:22     c_top_k_array_index
//   evaluation/backend/cython/include/holdout.h:20  evaluate_holdout
//   evaluation/backend/cython/include/loo.h:19      evaluate_loo
// Built only when /root/reference exists (oracle/Makefile target `ref`), output
// oracle/_ref/libref_eval.so (git-ignored AND gpurun-ignored: it exists in the build container only, the tests that
// use it skip on the GPU box).
// Used to pin oracle/mf_oracle.c (orc_topk / orc_holdout) and as the
// "reference" CPU baseline for top-k.
#include "func.h"
#include "holdout.h"
#include "loo.h"

extern "C" {

__attribute__((visibility("default")))
void ref_top_k_array_index(float *scores, int columns_num, int rows_num, int max_k, int *rankings)
{
    c_top_k_array_index(scores, columns_num, rows_num, max_k, rankings);
}

// ground truths as CSR; the int** table the reference wants is built here the
// way holdout_func.pyx:22-33 builds it (one pointer per user).
__attribute__((visibility("default")))
void ref_evaluate_holdout(int users_num, int *rankings, int max_k, int *Ks, int K_len,
                          const long long *t_indptr, int *t_indices, float *results)
{
    std::vector<int *> gt(users_num);
    std::vector<int> gt_num(users_num);
    for (int u = 0; u < users_num; ++u) {
        gt[u] = t_indices + t_indptr[u];
        gt_num[u] = (int)(t_indptr[u + 1] - t_indptr[u]);
    }
    evaluate_holdout(users_num, rankings, max_k, Ks, K_len, gt.data(), gt_num.data(), results);
}

// one held-out item per user; the int** table as loo_func.pyx builds it
__attribute__((visibility("default")))
void ref_evaluate_loo(int users_num, int *rankings, int max_k, int *Ks, int K_len, int *truth, float *results)
{
    std::vector<int *> gt(users_num);
    for (int u = 0; u < users_num; ++u) gt[u] = truth + u;
    evaluate_loo(users_num, rankings, max_k, Ks, K_len, gt.data(), results);
}

}
