/* oracle/asan_check.c -- TEST INFRASTRUCTURE: drives every oracle entry point once on small random
 * inputs; built with -fsanitize=address,undefined by `make -C oracle asan` (CPU only: GPU
 * AddressSanitizer is not available on the pool). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mf_oracle.c"

static float frand(void) { return (float)rand() / (float)RAND_MAX - 0.5f; }

int main(void)
{
    const int64_t U = 37, I = 23, B = 101;
    const int d = 32, K = 5;
    float *P = malloc(sizeof(float) * U * d), *Q = malloc(sizeof(float) * I * d);
    float *gP = calloc(U * d, sizeof(float)), *gQ = calloc(I * d, sizeof(float));
    float *mP = calloc(U * d, sizeof(float)), *vP = calloc(U * d, sizeof(float));
    float *mQ = calloc(I * d, sizeof(float)), *vQ = calloc(I * d, sizeof(float));
    int64_t *u = malloc(sizeof(int64_t) * B), *i = malloc(sizeof(int64_t) * B), *j = malloc(sizeof(int64_t) * B);
    for (int64_t n = 0; n < U * d; ++n) P[n] = 0.1f * frand();
    for (int64_t n = 0; n < I * d; ++n) Q[n] = 0.1f * frand();
    for (int64_t b = 0; b < B; ++b) { u[b] = rand() % U; i[b] = rand() % I; j[b] = rand() % I; }
    double loss = orc_bpr_loss(P, Q, u, i, j, B, d);
    orc_bpr_step_sgd(P, Q, U, I, u, i, j, B, d, 0.05f, gP, gQ, &loss);
    orc_bpr_step_adam(P, Q, U, I, u, i, j, B, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, mP, vP, mQ, vQ, gP, gQ, &loss);
    int64_t users[7] = {0, 5, 36, 1, 2, 3, 4};
    float *S = malloc(sizeof(float) * 7 * I);
    orc_score(P, users, 7, Q, I, d, S);
    int64_t indptr[38];
    int32_t indices[74];
    for (int r = 0; r <= 37; ++r) indptr[r] = 2 * r;
    for (int r = 0; r < 37; ++r) { indices[2 * r] = r % I; indices[2 * r + 1] = (r + 7) % I; }
    orc_mask_seen(S, users, 7, I, indptr, indices);
    int32_t *top = malloc(sizeof(int32_t) * 7 * K);
    orc_topk(S, I, 7, K, top);
    int32_t Ks[2] = {1, 5};
    float res[7 * 6];
    int64_t tp[8] = {0, 2, 4, 6, 8, 10, 12, 14};
    orc_holdout(7, top, K, Ks, 2, tp, indices, res);
    int32_t held[7] = {0, 3, 22, 1, 9, 4, 4};
    float res_loo[7 * 4];
    orc_loo(7, top, K, Ks, 2, held, res_loo);
    /* the pointwise branch: (user, item, rating) with repeats, both losses, both optimizers */
    float *y = malloc(sizeof(float) * B);
    for (int64_t b = 0; b < B; ++b) y[b] = (float)(rand() % 2);
    orc_pointwise_step_sgd(P, Q, U, I, u, i, y, B, d, 0, 0.05f, gP, gQ, &loss);
    orc_pointwise_step_sgd(P, Q, U, I, u, i, y, B, d, 1, 0.05f, gP, gQ, &loss);
    orc_pointwise_step_adam(P, Q, U, I, u, i, y, B, d, 0, 1e-3f, 0.9f, 0.999f, 1e-8f, 2, mP, vP, mQ, vQ, gP, gQ, &loss);
    free(y);
    /* LightGCN on a tiny symmetric graph: ring of N nodes */
    const int64_t N = U + I;
    int64_t *ap = malloc(sizeof(int64_t) * (N + 1));
    int32_t *ai = malloc(sizeof(int32_t) * 2 * N);
    float *av = malloc(sizeof(float) * 2 * N);
    for (int64_t r = 0; r <= N; ++r) ap[r] = 2 * r;
    for (int64_t r = 0; r < N; ++r) { ai[2 * r] = (int32_t)((r + N - 1) % N); ai[2 * r + 1] = (int32_t)((r + 1) % N); av[2 * r] = av[2 * r + 1] = 0.5f; }
    float *E0 = malloc(sizeof(float) * N * d), *m = calloc(N * d, sizeof(float)), *v = calloc(N * d, sizeof(float));
    float *s0 = malloc(sizeof(float) * N * d), *s1 = malloc(sizeof(float) * N * d), *s2 = malloc(sizeof(float) * N * d);
    float *s3 = malloc(sizeof(float) * N * d), *s4 = malloc(sizeof(float) * N * d);
    for (int64_t n = 0; n < N * d; ++n) E0[n] = 0.01f * frand();
    orc_lightgcn_step_adam(E0, m, v, U, I, ap, ai, av, 3, u, i, j, B, d, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, s0, s1, s2, s3, s4, &loss);
    printf("asan_check ok: loss %.6f top[0] %d ndcg %.4f\n", loss, top[0], res[5]);
    free(P); free(Q); free(gP); free(gQ); free(mP); free(vP); free(mQ); free(vQ); free(u); free(i); free(j);
    free(S); free(top); free(ap); free(ai); free(av); free(E0); free(m); free(v); free(s0); free(s1); free(s2); free(s3); free(s4);
    return 0;
}
