// rsx_bpr.hip -- BPR triplet step for gfx950 (MI355X): gather / dot / sigmoid /
// SGD write-back, item gradients by fp32 atomics, on-device triplet sampler.
//
// Arithmetic restated from the reference's eager path (no kernel exists there):
//   models/MF.py:32-42   r = sum(P[u] * Q[i])            (two gathers + dot)
//   models/MF.py:99-107  loss = -mean(log(sigmoid(r_pos - r_neg)))
//   models/MF.py:67-68   backward (dense grads, duplicates summed) + optimizer
// Closed form per triplet, all from PRE-step tables:
//   x = <P[u],Q[i]> - <P[u],Q[j]>;  g = dL/dx = -sigmoid(-x) / B
//   dP[u] += g (Q[i]-Q[j]);  dQ[i] += g P[u];  dQ[j] -= g P[u]
//
// Mapping to the machine: this is HBM/fabric-bound row traffic (24*d bytes per
// triplet algorithmically), no reuse, so no LDS staging and no MFMA.  A row of
// d fp32 is spread over LPR = d/4 lanes; a 64-lane wavefront therefore owns
// 64/LPR triplets at once (2 at d=128, 4 at d=64, 8 at d=32), the dot product
// is a butterfly inside the lane group, and every load/store/atomic of a row is
// coalesced over that lane group (see ROW LAYOUT below).
#include "rsx_common.h"

namespace {

constexpr int kBlock = 256;           // 4 wavefronts per workgroup
constexpr int kWavesPerBlock = kBlock / 64;

// ROW LAYOUT.  VEC=true : lane k holds elements [4k, 4k+4)  (one dwordx4 per row)
//              VEC=false: lane k holds elements k, k+LPR, k+2LPR, k+3LPR
//                         (four dword accesses, each LPR*4 contiguous bytes per row,
//                          so one atomic instruction touches ONE 128-B line per row
//                          at d=128 instead of four)
template <int D, bool VEC>
struct Row {
    static constexpr int LPR = D / 4;
    float v[4];
    __device__ __forceinline__ void load(const float *row, int k)
    {
        if constexpr (VEC) {
            float4 t = reinterpret_cast<const float4 *>(row)[k];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = row[k + c * LPR];
        }
    }
    // streaming (non-temporal) variants for rows nobody re-reads: keep them out of L2 / MALL
    __device__ __forceinline__ void load_nt(const float *row, int k)
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = VEC ? 4 * k + c : k + c * LPR;
            v[c] = __builtin_nontemporal_load(row + e);
        }
    }
    __device__ __forceinline__ void store_nt(float *row, int k) const
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = VEC ? 4 * k + c : k + c * LPR;
            __builtin_nontemporal_store(v[c], row + e);
        }
    }
    __device__ __forceinline__ void store(float *row, int k) const
    {
        if constexpr (VEC) {
            reinterpret_cast<float4 *>(row)[k] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) row[k + c * LPR] = v[c];
        }
    }
    // row[...] += s * v   (hardware fp32 atomics, no return)
    __device__ __forceinline__ void atomic_axpy(float *row, int k, float s) const
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = VEC ? 4 * k + c : k + c * LPR;
            rsx_atomic_add(row + e, s * v[c]);
        }
    }
};

template <int LPR>
__device__ __forceinline__ float group_sum(float x)
{
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// softplus(-x) = -log(sigmoid(x)), stable for any x (the reference's
// sigmoid().log() underflows for x < -88, MF.py:105; same value elsewhere)
__device__ __forceinline__ float softplus_neg(float x)
{
    return fmaxf(-x, 0.0f) + log1pf(__expf(-fabsf(x)));
}

// MODE 0: users unique in the batch -> P[u] updated in place by its owner group.
// MODE 1: users may repeat          -> user deltas summed into GU[owner slot].
template <int D, bool VEC, int MODE>
__global__ __launch_bounds__(kBlock) void bpr_step_kernel(
    float *__restrict__ P, const float *__restrict__ Q, float *__restrict__ G,
    const int32_t *__restrict__ U_idx, const int32_t *__restrict__ I_idx,
    const int32_t *__restrict__ J_idx, int64_t B, float lr, float inv_batch,
    float *__restrict__ loss_acc, const int32_t *__restrict__ owner, float *__restrict__ GU,
    int ablate)
{
    constexpr int LPR = D / 4;
    constexpr int TPW = 64 / LPR;  // triplets per wavefront
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

    float loss_local = 0.0f;
    int64_t b = wave * TPW + sub;
    const int64_t stride = nwaves * TPW;
    // indices of the first triplet (every lane of the group loads the same word)
    int32_t u = -1, i = -1, j = -1;
    if (b < B) { u = U_idx[b]; i = I_idx[b]; j = J_idx[b]; }
    while (b - sub < B) {   // wave-uniform trip count
        const int64_t bn = b + stride;
        int32_t un = -1, in = -1, jn = -1;
        if (bn < B) { un = U_idx[bn]; in = I_idx[bn]; jn = J_idx[bn]; }  // prefetch
        const bool live = (b < B) && (i >= 0);
        if (live) {
            float *prow = P + (size_t)u * D;
            const float *qi_row = Q + (size_t)i * D;
            const float *qj_row = Q + (size_t)j * D;
            Row<D, VEC> p, qi, qj;
            if (ablate & 8) p.load_nt(prow, k); else p.load(prow, k);
            qi.load(qi_row, k);
            qj.load(qj_row, k);
            float dpos = 0.0f, dneg = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dpos = fmaf(p.v[c], qi.v[c], dpos);
                dneg = fmaf(p.v[c], qj.v[c], dneg);
            }
            // (inactive groups skip the butterfly; the group is LPR-aligned so the
            //  xor partners are always inside the same live/dead group)
            dpos = group_sum<LPR>(dpos);
            dneg = group_sum<LPR>(dneg);
            const float x = dpos - dneg;
            const float sneg = 1.0f / (1.0f + __expf(x));      // sigmoid(-x)
            const float g = -sneg * inv_batch;                 // dL/dx
            if (k == 0) loss_local += softplus_neg(x);
            // item gradients (shared rows): G[i] += g p ; G[j] -= g p
            if (!(ablate & 1)) p.atomic_axpy(G + (size_t)i * D, k, g);
            if (!(ablate & 2)) p.atomic_axpy(G + (size_t)j * D, k, -g);
            // user row: P[u] -= lr * g * (qi - qj)
            const float s = -lr * g;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) p.v[c] = fmaf(s, qi.v[c] - qj.v[c], p.v[c]);
                if (!(ablate & 4)) { if (ablate & 16) p.store_nt(prow, k); else p.store(prow, k); }
            } else {
                Row<D, VEC> dq;
#pragma unroll
                for (int c = 0; c < 4; ++c) dq.v[c] = qi.v[c] - qj.v[c];
                const int32_t slot = owner[u] - 1;
                dq.atomic_axpy(GU + (size_t)slot * D, k, s);
            }
        }
        b = bn; u = un; i = in; j = jn;
    }
    if (loss_acc != nullptr) {
        const float w = wave_sum(loss_local);
        if (lane == 0) rsx_atomic_add(loss_acc + (wave & (RSX_LOSS_SLOTS - 1)), w);
    }
}

// claim: the first triplet (by CAS winner) of each distinct user owns that user's delta slot
__global__ __launch_bounds__(kBlock) void bpr_claim_kernel(const int32_t *__restrict__ U_idx,
                                                           const int32_t *__restrict__ I_idx,
                                                           int64_t B, int32_t *__restrict__ owner)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        if (I_idx[b] < 0) continue;
        atomicCAS(owner + U_idx[b], 0, (int32_t)(b + 1));
    }
}

// apply the summed user deltas: P[u] += GU[slot]; restore ws to all-zero
template <int D>
__global__ __launch_bounds__(kBlock) void bpr_apply_user_kernel(float *__restrict__ P,
                                                                const int32_t *__restrict__ U_idx,
                                                                const int32_t *__restrict__ I_idx,
                                                                int64_t B, int32_t *__restrict__ owner,
                                                                float *__restrict__ GU)
{
    constexpr int LPR = D / 4;
    constexpr int TPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * TPW;
    for (int64_t b = wave * TPW + sub; b < B; b += stride) {
        if (I_idx[b] < 0) continue;
        const int32_t u = U_idx[b];
        if (owner[u] != (int32_t)(b + 1)) continue;   // only the owner triplet applies
        float4 *gu = reinterpret_cast<float4 *>(GU + (size_t)b * D) + k;
        float4 *pr = reinterpret_cast<float4 *>(P + (size_t)u * D) + k;
        const float4 dlt = *gu;
        float4 p = *pr;
        p.x += dlt.x; p.y += dlt.y; p.z += dlt.z; p.w += dlt.w;
        *pr = p;
        *gu = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ __launch_bounds__(kBlock) void bpr_release_kernel(const int32_t *__restrict__ U_idx,
                                                             const int32_t *__restrict__ I_idx,
                                                             int64_t B, int32_t *__restrict__ owner)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        if (I_idx[b] < 0) continue;
        owner[U_idx[b]] = 0;   // benign same-value race between duplicates
    }
}

// r[b] = <P[u_b], Q[i_b]>  (MF.forward); one lane group per pair
template <int D>
__global__ __launch_bounds__(kBlock) void pair_score_kernel(const float *__restrict__ P,
                                                            const float *__restrict__ Q,
                                                            const int32_t *__restrict__ U_idx,
                                                            const int32_t *__restrict__ I_idx,
                                                            int64_t n, float *__restrict__ out)
{
    constexpr int LPR = D / 4;
    constexpr int TPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * TPW;
    for (int64_t b = wave * TPW + sub; b - sub < n; b += stride) {
        float acc = 0.0f;
        if (b < n) {
            Row<D, false> p, q;
            p.load(P + (size_t)U_idx[b] * D, k);
            q.load(Q + (size_t)I_idx[b] * D, k);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = fmaf(p.v[c], q.v[c], acc);
        }
        acc = group_sum<LPR>(acc);
        if (b < n && k == 0) out[b] = acc;
    }
}

// Q -= lr*G ; G = 0   (streaming; rows with an all-zero gradient quad are not written)
__global__ __launch_bounds__(kBlock) void apply_item_grad_kernel(float4 *__restrict__ Q,
                                                                 float4 *__restrict__ G, int64_t n4,
                                                                 float lr)
{
    for (int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x; n < n4; n += (int64_t)gridDim.x * kBlock) {
        const float4 g = G[n];
        if (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f) {
            float4 q = Q[n];
            q.x = fmaf(-lr, g.x, q.x); q.y = fmaf(-lr, g.y, q.y);
            q.z = fmaf(-lr, g.z, q.z); q.w = fmaf(-lr, g.w, q.w);
            Q[n] = q;
            G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// ---------------------------------------------------------------- sampler ------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t xorshift32(uint32_t &s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}

// keyed bijection of [0,n): 4-round Feistel over 2*hb bits with cycle walking
__device__ __forceinline__ uint32_t feistel_perm(uint32_t x, uint32_t n, int hb, uint64_t key)
{
    const uint32_t mask = (1u << hb) - 1u;
    do {
        uint32_t l = x >> hb, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = (uint32_t)splitmix64(key ^ ((uint64_t)r << 8) ^ (uint64_t)round) & mask;
            const uint32_t t = l ^ f;
            l = r; r = t;
        }
        x = (l << hb) | r;
    } while (x >= n);
    return x;
}

__global__ __launch_bounds__(kBlock) void bpr_sample_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t U, int64_t I,
    int64_t B, uint64_t seed, uint64_t step, int64_t epoch_pos, int hb,
    int32_t *__restrict__ u_out, int32_t *__restrict__ i_out, int32_t *__restrict__ j_out)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        const int64_t gpos = epoch_pos + b;
        const uint64_t epoch = (uint64_t)(gpos / U);
        const uint32_t pos = (uint32_t)(gpos % U);
        const uint32_t u = feistel_perm(pos, (uint32_t)U, hb, splitmix64(seed ^ (epoch * 0xD1B54A32D192ED03ull)));
        uint32_t s = (uint32_t)splitmix64(seed ^ (step * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)b * 0xBF58476D1CE4E5B9ull));
        s |= (s == 0);
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        const uint32_t deg = (uint32_t)(hi - lo);
        int32_t pi = -1, nj = -1;
        if (deg > 0 && (int64_t)deg < I) {
            pi = indices[lo + (int64_t)(((uint64_t)xorshift32(s) * deg) >> 32)];
            for (;;) {
                nj = (int32_t)(((uint64_t)xorshift32(s) * (uint64_t)I) >> 32);
                int64_t a = lo, z = hi;     // binary search in the sorted row
                while (a < z) {
                    const int64_t m = (a + z) >> 1;
                    if (indices[m] < nj) a = m + 1; else z = m;
                }
                if (!(a < hi && indices[a] == nj)) break;
            }
        }
        u_out[b] = (int32_t)u; i_out[b] = pi; j_out[b] = nj;
    }
}

int g_layout_vec = 0;   // row layout used by bpr_step (0 = strided dwords, 1 = dwordx4); tuning knob
int g_ablate = 0;       // development only: 1 = skip pos-item atomics, 2 = skip neg-item atomics, 4 = skip P store

template <int D, bool VEC, int MODE>
void launch_step(float *P, const float *Q, float *G, const int32_t *u, const int32_t *i,
                 const int32_t *j, int64_t B, float lr, float inv_batch, float *loss_acc,
                 const int32_t *owner, float *GU, int ablate, hipStream_t st)
{
    constexpr int TPW = 64 / (D / 4);
    const int64_t waves = (B + TPW - 1) / TPW;
    int64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = (int64_t)rsx_num_cus() * 8;   // 8 blocks x 4 waves = 32 waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((bpr_step_kernel<D, VEC, MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, st, P, Q,
                       G, u, i, j, B, lr, inv_batch, loss_acc, owner, GU, ablate);
}

template <int MODE>
void dispatch_step(int d, bool vec, float *P, const float *Q, float *G, const int32_t *u,
                   const int32_t *i, const int32_t *j, int64_t B, float lr, float inv_batch,
                   float *loss_acc, const int32_t *owner, float *GU, int ablate, hipStream_t st)
{
#define RSX_CASE(DD)                                                                                   \
    case DD:                                                                                           \
        if (vec) launch_step<DD, true, MODE>(P, Q, G, u, i, j, B, lr, inv_batch, loss_acc, owner, GU, ablate, st); \
        else launch_step<DD, false, MODE>(P, Q, G, u, i, j, B, lr, inv_batch, loss_acc, owner, GU, ablate, st);    \
        break;
    switch (d) { RSX_CASE(32) RSX_CASE(64) RSX_CASE(128) }
#undef RSX_CASE
}

int64_t grid_1d(int64_t n)
{
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rsx_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

}  // namespace

// undocumented tuning hook (bench / tests): select the row layout of bpr_step
RSX_API int rsx_debug_set_layout(int vec) { g_layout_vec = vec ? 1 : 0; return RSX_OK; }
RSX_API int rsx_debug_set_ablation(int mask) { g_ablate = mask; return RSX_OK; }

RSX_API int64_t rsx_bpr_step_workspace(int64_t num_users, int64_t max_batch, int d)
{
    if (num_users < 0 || max_batch < 0 || !rsx_dim_ok(d)) return RSX_E_INVALID;
    const int64_t owner_bytes = ((num_users * 4 + 255) / 256) * 256;
    return owner_bytes + max_batch * d * 4;
}

RSX_API int rsx_bpr_step(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items,
                         const int32_t *u_dev, const int32_t *i_dev, const int32_t *j_dev,
                         int64_t batch, int d, float lr, float inv_batch, float *loss_acc,
                         unsigned flags, void *ws, int64_t ws_bytes, rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && (G || (flags & RSX_NO_UPDATE)), "null table pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64 or 128");
    RSX_CHECK_ARG(batch >= 0 && num_users > 0 && num_items > 0, "negative size");
    if (batch == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev && j_dev, "null index pointer");
    hipStream_t st = (hipStream_t)stream;
    if (flags & RSX_NO_UPDATE) {   // loss only: the in-place kernel with every write suppressed
        dispatch_step<0>(d, g_layout_vec != 0, P, Q, G, u_dev, i_dev, j_dev, batch, lr, inv_batch,
                         loss_acc, nullptr, nullptr, /*suppress every write*/ 7, st);
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    if (flags & RSX_USERS_UNIQUE) {
        dispatch_step<0>(d, g_layout_vec != 0, P, Q, G, u_dev, i_dev, j_dev, batch, lr, inv_batch,
                         loss_acc, nullptr, nullptr, g_ablate, st);
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    const int64_t need = rsx_bpr_step_workspace(num_users, batch, d);
    if (ws == nullptr || ws_bytes < need) {
        rsx_set_error("rsx_bpr_step: workspace of %lld bytes required without RSX_USERS_UNIQUE, got %lld",
                      (long long)need, (long long)ws_bytes);
        return RSX_E_WORKSPACE;
    }
    int32_t *owner = (int32_t *)ws;
    float *GU = (float *)((char *)ws + ((num_users * 4 + 255) / 256) * 256);
    const unsigned g1 = (unsigned)grid_1d(batch);
    hipLaunchKernelGGL(bpr_claim_kernel, dim3(g1), dim3(kBlock), 0, st, u_dev, i_dev, batch, owner);
    dispatch_step<1>(d, g_layout_vec != 0, P, Q, G, u_dev, i_dev, j_dev, batch, lr, inv_batch,
                     loss_acc, owner, GU, g_ablate, st);
    const int64_t tpw = 64 / (d / 4);
    const unsigned g2 = (unsigned)grid_1d((batch + tpw - 1) / tpw * 64);
    switch (d) {
    case 32: hipLaunchKernelGGL(bpr_apply_user_kernel<32>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    case 64: hipLaunchKernelGGL(bpr_apply_user_kernel<64>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    default: hipLaunchKernelGGL(bpr_apply_user_kernel<128>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    }
    hipLaunchKernelGGL(bpr_release_kernel, dim3(g1), dim3(kBlock), 0, st, u_dev, i_dev, batch, owner);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_pair_score(const float *P, const float *Q, const int32_t *u_dev, const int32_t *i_dev,
                           int64_t n, int d, float *r_out, rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && r_out, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && n >= 0, "bad shape");
    if (n == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev, "null index pointer");
    const int64_t tpw = 64 / (d / 4);
    const unsigned g = (unsigned)grid_1d((n + tpw - 1) / tpw * 64);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
    case 32: hipLaunchKernelGGL(pair_score_kernel<32>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    case 64: hipLaunchKernelGGL(pair_score_kernel<64>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    default: hipLaunchKernelGGL(pair_score_kernel<128>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_apply_item_grad(float *Q, float *G, int64_t num_items, int d, float lr,
                                rsx_stream_t stream)
{
    RSX_CHECK_ARG(Q && G, "null table pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_items > 0, "bad shape");
    const int64_t n4 = num_items * d / 4;
    hipLaunchKernelGGL(apply_item_grad_kernel, dim3((unsigned)grid_1d(n4)), dim3(kBlock), 0,
                       (hipStream_t)stream, (float4 *)Q, (float4 *)G, n4, lr);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_bpr_sample(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                           int64_t num_items, int64_t batch, uint64_t seed, uint64_t step,
                           int64_t epoch_pos, int32_t *u_out, int32_t *i_out, int32_t *j_out,
                           rsx_stream_t stream)
{
    RSX_CHECK_ARG(indptr_dev && indices_dev && u_out && i_out && j_out, "null pointer");
    RSX_CHECK_ARG(num_users > 0 && num_users < (1ll << 31) && num_items > 0 && num_items < (1ll << 31),
                  "table sizes must fit int32");
    RSX_CHECK_ARG(batch >= 0 && epoch_pos >= 0, "negative size");
    if (batch == 0) return RSX_OK;
    int bits = 1;
    while ((1ll << bits) < num_users) ++bits;
    const int hb = (bits + 1) / 2;
    hipLaunchKernelGGL(bpr_sample_kernel, dim3((unsigned)grid_1d(batch)), dim3(kBlock), 0,
                       (hipStream_t)stream, indptr_dev, indices_dev, num_users, num_items, batch, seed,
                       step, epoch_pos, hb, u_out, i_out, j_out);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}
