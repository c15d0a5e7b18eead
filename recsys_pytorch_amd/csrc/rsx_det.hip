// rsx_det.hip -- the DETERMINISTIC form of the BPR step (rsx_bpr_step with RSX_DETERMINISTIC).
//
// The production step kernels sum item gradients with fp32 atomics, whose order differs from run
// to run: results agree to rounding, not bit for bit.  For bisecting ("hard
// parts") this file computes the SAME step -- models/MF.py:64-68 with SGD, every gradient taken at
// the pre-step tables, duplicates summed -- without a single atomic and in one fixed order, the
// order autograd's index_add uses on the CPU: for every item row, its incidences in ascending
// batch position.
//   1. coefficients   g_b = -sigmoid(-x_b) / B and softplus(-x_b) per triplet, plus the 2B (item,
//                     position) incidences:  (i_b, 2b) for the positive, (j_b, 2b+1) for the negative
//   2. order          stable device radix sort of the incidences by item (rocPRIM, a plain library
//                     primitive): inside an item they stay in ascending position
//   3. item rows      one lane group per item walks its incidences in that order,
//                     acc += (+/-) g_b * P[u_b] (P still pre-step), and writes G[item] += acc once
//   4. user rows      P[u_b] -= lr * g_b * (Q[i_b] - Q[j_b]) in place (users unique in the batch)
//   5. loss           the per-triplet terms summed by ONE workgroup in a fixed tree order
// Debugging aid, not a fast path: a popular item's whole run is one lane group's serial loop.
#include <rocprim/device/device_radix_sort.hpp>

#include "rsx_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int LPR = 32;       // lanes per row (same row layout as rsx_bpr.hip)

__device__ __forceinline__ float lane_group_sum(float x)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);   // stays inside the 32-lane group
    return x;
}

__device__ __forceinline__ float softplus_neg(float x) { return fmaxf(-x, 0.0f) + log1pf(__expf(-fabsf(x))); }

template <int D>
__global__ __launch_bounds__(kBlock) void det_coef_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                          const int32_t *__restrict__ U_idx, const int32_t *__restrict__ I_idx,
                                                          const int32_t *__restrict__ J_idx, int64_t B, int32_t num_items,
                                                          float inv_batch, float *__restrict__ coef, float *__restrict__ lossv,
                                                          uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    constexpr int EPL = D / 32;
    const int lane = threadIdx.x & 63, sub = lane / LPR, k = lane % LPR;
    const int64_t group = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + sub;
    const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * 2;
    for (int64_t b = group; b - sub < B; b += stride) {       // wave-uniform trip count
        const bool in = b < B;
        const int32_t i = in ? I_idx[b] : -1;
        float x = 0.f;
        if (in && i >= 0) {
            const float *p = P + (size_t)U_idx[b] * D, *qi = Q + (size_t)i * D, *qj = Q + (size_t)J_idx[b] * D;
            float dpos = 0.f, dneg = 0.f;
#pragma unroll
            for (int c = 0; c < EPL; ++c) {
                const float pv = p[k + 32 * c];
                dpos = fmaf(pv, qi[k + 32 * c], dpos);
                dneg = fmaf(pv, qj[k + 32 * c], dneg);
            }
            x = lane_group_sum(dpos) - lane_group_sum(dneg);
        } else {
            (void)lane_group_sum(0.f); (void)lane_group_sum(0.f);
        }
        if (in && k == 0) {
            const bool live = i >= 0;
            coef[b] = live ? -(1.0f / (1.0f + __expf(x))) * inv_batch : 0.f;
            lossv[b] = live ? softplus_neg(x) : 0.f;
            keys[2 * b] = live ? (uint32_t)i : (uint32_t)num_items;            // skipped triplets sort last
            keys[2 * b + 1] = live ? (uint32_t)J_idx[b] : (uint32_t)num_items;
            vals[2 * b] = (uint32_t)(2 * b);
            vals[2 * b + 1] = (uint32_t)(2 * b + 1);
        }
    }
}

// one lane group per segment head of the sorted incidences
template <int D>
__global__ __launch_bounds__(kBlock) void det_item_kernel(const float *__restrict__ P, float *__restrict__ G,
                                                          const int32_t *__restrict__ U_idx, const float *__restrict__ coef,
                                                          const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                          int64_t n_inc, int32_t num_items)
{
    constexpr int EPL = D / 32;
    const int lane = threadIdx.x & 63, sub = lane / LPR, k = lane % LPR;
    const int64_t group = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + sub;
    const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * 2;
    for (int64_t p = group; p < n_inc; p += stride) {
        const uint32_t item = keys[p];
        if (item >= (uint32_t)num_items || (p > 0 && keys[p - 1] == item)) continue;       // not a segment head
        float acc[EPL];
#pragma unroll
        for (int c = 0; c < EPL; ++c) acc[c] = 0.f;
        for (int64_t q = p; q < n_inc && keys[q] == item; ++q) {           // ascending batch position
            const uint32_t v = vals[q];
            const int64_t b = v >> 1;
            const float g = (v & 1u) ? -coef[b] : coef[b];                // G[i] += g p ; G[j] -= g p
            const float *prow = P + (size_t)U_idx[b] * D;
#pragma unroll
            for (int c = 0; c < EPL; ++c) acc[c] = fmaf(g, prow[k + 32 * c], acc[c]);
        }
        float *grow = G + (size_t)item * D;
#pragma unroll
        for (int c = 0; c < EPL; ++c) grow[k + 32 * c] += acc[c];          // this lane group owns the row
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void det_user_kernel(float *__restrict__ P, const float *__restrict__ Q,
                                                          const int32_t *__restrict__ U_idx, const int32_t *__restrict__ I_idx,
                                                          const int32_t *__restrict__ J_idx, const float *__restrict__ coef,
                                                          int64_t B, float lr)
{
    constexpr int EPL = D / 32;
    const int lane = threadIdx.x & 63, sub = lane / LPR, k = lane % LPR;
    const int64_t group = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 2 + sub;
    const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * 2;
    for (int64_t b = group; b < B; b += stride) {
        const int32_t i = I_idx[b];
        if (i < 0) continue;
        const float s = -lr * coef[b];
        float *p = P + (size_t)U_idx[b] * D;
        const float *qi = Q + (size_t)i * D, *qj = Q + (size_t)J_idx[b] * D;
#pragma unroll
        for (int c = 0; c < EPL; ++c) p[k + 32 * c] = fmaf(s, qi[k + 32 * c] - qj[k + 32 * c], p[k + 32 * c]);
    }
}

// fixed-order sum of n floats by one workgroup: thread t sums elements t, t+256, ... in order, then a tree
__global__ __launch_bounds__(kBlock) void det_sum_kernel(const float *__restrict__ v, int64_t n, float *__restrict__ out)
{
    __shared__ float part[kBlock];
    float a = 0.f;
    for (int64_t q = threadIdx.x; q < n; q += kBlock) a += v[q];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int w = kBlock / 2; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += part[0];
}

int64_t a256(int64_t x) { return (x + 255) / 256 * 256; }

int bits_for(int64_t num_items)   // keys are in [0, num_items]
{
    int bits = 1;
    while ((1ll << bits) <= num_items) ++bits;
    return bits;
}

size_t sort_temp(int64_t n, int bits)
{
    size_t bytes = 0;
    uint32_t *nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)n, 0, (unsigned)bits, (hipStream_t)0);
    return bytes;
}

unsigned grid_for(int64_t groups)
{
    int64_t blocks = (groups + 7) / 8;            // 8 lane groups per 256-thread block
    const int64_t cap = (int64_t)rsx_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

RSX_API int64_t rsx_bpr_step_det_workspace(int64_t batch, int64_t num_items)
{
    if (batch < 0 || num_items <= 0 || num_items >= (1ll << 31) || batch >= (1ll << 30)) return RSX_E_INVALID;
    if (batch == 0) return 0;
    return 2 * a256(batch * 4) + 4 * a256(2 * batch * 4) + a256((int64_t)sort_temp(2 * batch, bits_for(num_items)));
}

// called by rsx_bpr_step (rsx_bpr.hip) when RSX_DETERMINISTIC is set; arguments already validated
int rsx_bpr_step_deterministic(float *P, const float *Q, float *G, int64_t num_items, const int32_t *u_dev,
                               const int32_t *i_dev, const int32_t *j_dev, int64_t batch, int d, float lr, float inv_batch,
                               float *loss_acc, void *ws, int64_t ws_bytes, hipStream_t st)
{
    const int64_t need = rsx_bpr_step_det_workspace(batch, num_items);
    if (need < 0) { rsx_set_error("rsx_bpr_step: RSX_DETERMINISTIC needs batch < 2^30"); return RSX_E_INVALID; }
    if (ws == nullptr || ws_bytes < need) {
        rsx_set_error("rsx_bpr_step: RSX_DETERMINISTIC needs a workspace of %lld bytes (rsx_bpr_step_det_workspace), got %lld",
                      (long long)need, (long long)ws_bytes);
        return RSX_E_WORKSPACE;
    }
    char *w = (char *)ws;
    float *coef = (float *)w;            w += a256(batch * 4);
    float *lossv = (float *)w;           w += a256(batch * 4);
    uint32_t *keys_in = (uint32_t *)w;   w += a256(2 * batch * 4);
    uint32_t *vals_in = (uint32_t *)w;   w += a256(2 * batch * 4);
    uint32_t *keys_out = (uint32_t *)w;  w += a256(2 * batch * 4);
    uint32_t *vals_out = (uint32_t *)w;  w += a256(2 * batch * 4);
    const int bits = bits_for(num_items);
    size_t temp_bytes = sort_temp(2 * batch, bits);
    const unsigned g1 = grid_for(batch), g2 = grid_for(2 * batch);
#define RSX_DET(D_)                                                                                                          \
    hipLaunchKernelGGL(det_coef_kernel<D_>, dim3(g1), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, j_dev, batch, (int32_t)num_items,  \
                       inv_batch, coef, lossv, keys_in, vals_in);                                                            \
    if (rocprim::radix_sort_pairs(w, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)(2 * batch), 0, (unsigned)bits, \
                                  st) != hipSuccess) { rsx_set_error("rsx_bpr_step: radix sort failed"); return RSX_E_HIP; } \
    hipLaunchKernelGGL(det_item_kernel<D_>, dim3(g2), dim3(kBlock), 0, st, P, G, u_dev, coef, keys_out, vals_out, 2 * batch,  \
                       (int32_t)num_items);                                                                                  \
    hipLaunchKernelGGL(det_user_kernel<D_>, dim3(g1), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, j_dev, coef, batch, lr);
    switch (d) {
    case 32: RSX_DET(32) break;
    case 64: RSX_DET(64) break;
    case 128: RSX_DET(128) break;
    default: RSX_DET(256) break;
    }
#undef RSX_DET
    if (loss_acc != nullptr) hipLaunchKernelGGL(det_sum_kernel, dim3(1), dim3(kBlock), 0, st, lossv, batch, loss_acc);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}
