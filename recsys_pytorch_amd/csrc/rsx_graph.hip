// rsx_graph.hip -- LightGCN embedding propagation for gfx950 (SURVEY section 8f row f1,
// BASELINE config 5): Y = A_hat X as a row gather, the same access pattern as bpr_step's gather.
//
// Restates models/LightGCN.py:188-197: `all_emb = torch.sparse.mm(graph, all_emb)` with
// A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 (built on the host exactly as the reference does,
// LightGCN.py:228-258, and handed over as CSR).  A_hat is symmetric, so the backward pass of
// the propagation is the same product applied to the gradient.
//
// HBM-bound: nnz * (4 d + 8) bytes gathered per product.  Item rows of a popularity-skewed
// graph are enormous (hundreds of thousands of neighbours), user rows tiny: the rows are cut
// into segments of at most `seg` non-zeros by a host-side plan (the graph is static), one lane
// group per segment; a row with a single segment is stored, a split row is combined with fp32
// atomics into a pre-zeroed Y.
#include <vector>

#include "rsx_common.h"

namespace {

constexpr int kBlock = 256;

// what a flagged-off neighbour reads instead of its row: ONE shared all-zero row (always an L1 / L2 hit).  Selecting the ADDRESS
// keeps every fetch an unconditional dwordx4; selecting the VALUE (`flag ? load : zero`) made the compiler split each fetch into
// four exec-masked dword loads behind their own branches, which cost what the skipped rows saved (round 3, measured: no gain).
__device__ __attribute__((aligned(16))) float g_zero_row[256];

// SKIP: `nz` flags the rows of X that are not entirely zero; a flagged-off neighbour row is not fetched (a * 0 adds nothing:
// the result is bit-identical).  The first backward product of a LightGCN step multiplies A_hat with the dense gradient of the
// loss, of which only the batch's users' and items' rows are non-zero: with 65 536 of 1M users in the batch, 93 % of the user
// rows an item row would gather are zeros (models/LightGCN.py:83-87 back-propagates through the same dense product).
#ifndef RSX_SPMM_BLOCKS_PER_CU
#define RSX_SPMM_BLOCKS_PER_CU 1024  // grid cap of the product (workgroups of 4 wavefronts per CU): ms per product at 4 / 8 / 16 / 64 / 256 / 1024 per CU: 2.51 / 2.50 / 2.48 / 2.44 / 2.42 / 2.40 -- short workgroups the hardware deals out beat a fixed stride over the segments
#endif
#ifndef RSX_SPMM_PIPELINE
#define RSX_SPMM_PIPELINE 1      // 0: the round-2 loop (development A/B: 2.59 -> 2.53 ms per product at the configs[4] shape)
#endif
template <int D, bool SKIP>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(
    const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_begin,
    const int32_t *__restrict__ seg_len, int64_t num_segs, const int64_t *__restrict__ indptr,
    const int32_t *__restrict__ indices, const float *__restrict__ vals, const float *__restrict__ X,
    float *__restrict__ Y, float *__restrict__ S, const uint8_t *__restrict__ nz, const uint8_t *__restrict__ want,
    const float *__restrict__ Sinit)
{
    constexpr int LPR = D / 4;
    constexpr int GPW = 64 / LPR;                 // lane groups (segments) per wavefront
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t s = wave * GPW + sub; s < num_segs; s += nwaves * GPW) {
        const int32_t row = seg_row[s];
        // `want` (nullable): only the flagged rows of Y are computed -- the LAST forward product of a LightGCN training step, whose
        // result is read at the batch's users and items only (models/LightGCN.py:117-123 indexes the propagated tables by the batch)
        if (want != nullptr && want[row] == 0) continue;
        if (RSX_ABL(1) && s == 1) continue;       // (dev build only: one segment of the product dropped, tests/test_mutation.py)
        const int64_t pb = seg_begin[s];
        const int len = seg_len[s];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int p = 0;
#if RSX_SPMM_PIPELINE
        // four neighbour rows in flight, and the NEXT four (value, index) pairs requested before this group's rows are used: one
        // dependent round trip per group instead of two
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        if (len >= 4) {
            a0 = vals[pb]; a1 = vals[pb + 1]; a2 = vals[pb + 2]; a3 = vals[pb + 3];
            n0 = indices[pb]; n1 = indices[pb + 1]; n2 = indices[pb + 2]; n3 = indices[pb + 3];
        }
        for (; p + 4 <= len; p += 4) {
            const float *r0 = X + (size_t)n0 * D, *r1 = X + (size_t)n1 * D, *r2 = X + (size_t)n2 * D, *r3 = X + (size_t)n3 * D;
            if constexpr (SKIP) {
                const uint8_t f0 = nz[n0], f1 = nz[n1], f2 = nz[n2], f3 = nz[n3];
                r0 = f0 ? r0 : g_zero_row; r1 = f1 ? r1 : g_zero_row; r2 = f2 ? r2 : g_zero_row; r3 = f3 ? r3 : g_zero_row;
            }
            const float4 x0 = reinterpret_cast<const float4 *>(r0)[k];
            const float4 x1 = reinterpret_cast<const float4 *>(r1)[k];
            const float4 x2 = reinterpret_cast<const float4 *>(r2)[k];
            const float4 x3 = reinterpret_cast<const float4 *>(r3)[k];
            const float c0 = a0, c1 = a1, c2 = a2, c3 = a3;
            if (p + 8 <= len) {
                a0 = vals[pb + p + 4]; a1 = vals[pb + p + 5]; a2 = vals[pb + p + 6]; a3 = vals[pb + p + 7];
                n0 = indices[pb + p + 4]; n1 = indices[pb + p + 5]; n2 = indices[pb + p + 6]; n3 = indices[pb + p + 7];
            }
            acc.x = fmaf(c0, x0.x, acc.x); acc.y = fmaf(c0, x0.y, acc.y); acc.z = fmaf(c0, x0.z, acc.z); acc.w = fmaf(c0, x0.w, acc.w);
            acc.x = fmaf(c1, x1.x, acc.x); acc.y = fmaf(c1, x1.y, acc.y); acc.z = fmaf(c1, x1.z, acc.z); acc.w = fmaf(c1, x1.w, acc.w);
            acc.x = fmaf(c2, x2.x, acc.x); acc.y = fmaf(c2, x2.y, acc.y); acc.z = fmaf(c2, x2.z, acc.z); acc.w = fmaf(c2, x2.w, acc.w);
            acc.x = fmaf(c3, x3.x, acc.x); acc.y = fmaf(c3, x3.y, acc.y); acc.z = fmaf(c3, x3.z, acc.z); acc.w = fmaf(c3, x3.w, acc.w);
        }
#else
        for (; p + 4 <= len; p += 4) {            // four neighbour rows in flight
            const float a0 = vals[pb + p], a1 = vals[pb + p + 1], a2 = vals[pb + p + 2], a3 = vals[pb + p + 3];
            const int32_t n0 = indices[pb + p], n1 = indices[pb + p + 1], n2 = indices[pb + p + 2], n3 = indices[pb + p + 3];
            const float *r0 = X + (size_t)n0 * D, *r1 = X + (size_t)n1 * D, *r2 = X + (size_t)n2 * D, *r3 = X + (size_t)n3 * D;
            if constexpr (SKIP) {
                const uint8_t f0 = nz[n0], f1 = nz[n1], f2 = nz[n2], f3 = nz[n3];
                r0 = f0 ? r0 : g_zero_row; r1 = f1 ? r1 : g_zero_row; r2 = f2 ? r2 : g_zero_row; r3 = f3 ? r3 : g_zero_row;
            }
            const float4 x0 = reinterpret_cast<const float4 *>(r0)[k];
            const float4 x1 = reinterpret_cast<const float4 *>(r1)[k];
            const float4 x2 = reinterpret_cast<const float4 *>(r2)[k];
            const float4 x3 = reinterpret_cast<const float4 *>(r3)[k];
            acc.x = fmaf(a0, x0.x, acc.x); acc.y = fmaf(a0, x0.y, acc.y); acc.z = fmaf(a0, x0.z, acc.z); acc.w = fmaf(a0, x0.w, acc.w);
            acc.x = fmaf(a1, x1.x, acc.x); acc.y = fmaf(a1, x1.y, acc.y); acc.z = fmaf(a1, x1.z, acc.z); acc.w = fmaf(a1, x1.w, acc.w);
            acc.x = fmaf(a2, x2.x, acc.x); acc.y = fmaf(a2, x2.y, acc.y); acc.z = fmaf(a2, x2.z, acc.z); acc.w = fmaf(a2, x2.w, acc.w);
            acc.x = fmaf(a3, x3.x, acc.x); acc.y = fmaf(a3, x3.y, acc.y); acc.z = fmaf(a3, x3.z, acc.z); acc.w = fmaf(a3, x3.w, acc.w);
        }
#endif
        for (; p < len; ++p) {
            const float a = vals[pb + p];
            const int32_t n = indices[pb + p];
            const float *r = X + (size_t)n * D;
            if constexpr (SKIP) r = nz[n] ? r : g_zero_row;
            const float4 x = reinterpret_cast<const float4 *>(r)[k];
            acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y); acc.z = fmaf(a, x.z, acc.z); acc.w = fmaf(a, x.w, acc.w);
        }
        const bool whole = (int64_t)len == indptr[row + 1] - indptr[row];
        float *y = Y + (size_t)row * D + 4 * k;
        if (whole) {
            *reinterpret_cast<float4 *>(y) = acc;
            if (S != nullptr) {
                // S_init (nullable): S = S_init + A X instead of S += A X -- the first product of a propagation, whose running layer
                // sum starts as the source table itself (saves the copy of the table into S beforehand)
                float4 *sp = reinterpret_cast<float4 *>(S + (size_t)row * D) + k;
                float4 t = Sinit != nullptr ? reinterpret_cast<const float4 *>(Sinit + (size_t)row * D)[k] : *sp;
                t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
                *sp = t;
            }
        } else {
            rsx_atomic_add(y, acc.x); rsx_atomic_add(y + 1, acc.y); rsx_atomic_add(y + 2, acc.z); rsx_atomic_add(y + 3, acc.w);
            if (S != nullptr) {
                float *sp = S + (size_t)row * D + 4 * k;
                rsx_atomic_add(sp, acc.x); rsx_atomic_add(sp + 1, acc.y); rsx_atomic_add(sp + 2, acc.z); rsx_atomic_add(sp + 3, acc.w);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void scale_kernel(float4 *__restrict__ X, int64_t n4, float alpha)
{
    for (int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x; n < n4; n += (int64_t)gridDim.x * kBlock) {
        float4 v = X[n];
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        X[n] = v;
    }
}

// rows cut into several segments are summed with atomics into zeros: only THEIR rows of Y are cleared (a memset of all of Y -- 563 MB at
// the configs[4] shape -- for a few thousand rows cost 3 % of a product).  The first segment of a split row clears it.
template <int D>
__global__ __launch_bounds__(kBlock) void zero_split_rows_kernel(const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_begin,
                                                                 const int32_t *__restrict__ seg_len, int64_t num_segs,
                                                                 const int64_t *__restrict__ indptr, float *__restrict__ Y,
                                                                 float *__restrict__ S, const float *__restrict__ Sinit,
                                                                 const uint8_t *__restrict__ want)
{
    // one thread per segment decides; the few that start a split row clear it (a thread per quad of every segment -- 35M threads
    // at the configs[4] shape -- took 99 us, more than the memset it had replaced)
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= num_segs) return;
    const int32_t row = seg_row[s];
    if (want != nullptr && want[row] == 0) return;       // rsx_spmm_csr_select_rows: an unwanted row is NOT written, split or not
    const int64_t lo = indptr[row], hi = indptr[row + 1];
    if (seg_begin[s] == lo && (int64_t)seg_len[s] < hi - lo) {
        float4 *y = reinterpret_cast<float4 *>(Y + (size_t)row * D);
#pragma unroll 8
        for (int k = 0; k < D / 4; ++k) y[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (S != nullptr && Sinit != nullptr) {      // S = S_init + A X: the split row's pieces add atomically into the initial row
            float4 *sp = reinterpret_cast<float4 *>(S + (size_t)row * D);
            const float4 *ip = reinterpret_cast<const float4 *>(Sinit + (size_t)row * D);
#pragma unroll 8
            for (int k = 0; k < D / 4; ++k) sp[k] = ip[k];
        }
    }
}

// ---- the longest rows by SCATTER: each source row read once -----------------------------------------
// A popularity-skewed interaction graph puts half of all non-zeros into a few hundred item rows (Zipf, 100 000 items: the 128 longest rows
// hold 45 % of the entries).  As a gather those rows re-read the user table once per row: every user row is fetched ~10 times for them
// (5 GB of the product's 10.6 GB at the configs[4] shape).  Turned round -- walk the SOURCE rows in order, read each once, and add a x into
// the accumulators of the hot rows it belongs to -- the same entries cost one streaming pass over the source rows that have a hot
// neighbour (0.5 GB).  The accumulators of H = 16 384 / d hot rows live in LDS (64 KB per workgroup, ds_add_f32), one workgroup per CU;
// at the end every workgroup adds its partial rows to Y (pre-zeroed) with global atomics -- H x workgroups row updates, not one per entry.
// The caller's plan (rsx_spmm_plan) simply owns no segment for the hot rows; rsx_spmm_hot_rows computes them.
constexpr int kHotThreads = 1024;
constexpr int kHotLdsFloats = 16384;          // 64 KB of accumulators: H * D <= 16 384

template <int D>
__global__ __launch_bounds__(kHotThreads) void spmm_hot_rows_kernel(rsx_spmm_hot h, const float *__restrict__ X, const uint8_t *__restrict__ nz,
                                                                    const uint8_t *__restrict__ want, float *__restrict__ Y)
{
    __shared__ __attribute__((aligned(16))) float acc[kHotLdsFloats];
    constexpr int LPR = D / 4;                 // lanes per row (a float4 each)
    constexpr int GROUPS = kHotThreads / LPR;
    const int tid = threadIdx.x, g = tid / LPR, k = tid % LPR;
    const int H = h.num_hot;
    for (int q = tid; q < H * D; q += kHotThreads) acc[q] = 0.f;
    __syncthreads();
    // a contiguous share of the source rows per workgroup (they ascend: the source table is streamed), the groups side by side inside it
    const int64_t per = ceil_div64(h.num_src, gridDim.x), s0 = (int64_t)blockIdx.x * per, s1 = (s0 + per < h.num_src) ? s0 + per : h.num_src;
    for (int64_t s = s0 + g; s < s1; s += GROUPS) {
        const int32_t c = h.src_rows[s];
        if (nz != nullptr && nz[c] == 0) continue;                  // (a * 0 adds nothing: the row is not fetched)
        const float4 x = reinterpret_cast<const float4 *>(X + (size_t)c * D)[k];
        const int64_t e0 = h.src_ptr[s], e1 = h.src_ptr[s + 1];
        for (int64_t e = e0; e < e1; ++e) {
            const float a = h.src_val[e];
            float *dst = acc + (int)h.src_slot[e] * D + 4 * k;
            __hip_atomic_fetch_add(dst, a * x.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 1, a * x.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 2, a * x.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 3, a * x.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (s0 >= s1) return;                                           // (a workgroup without a share adds nothing)
    for (int q = tid; q < H * D; q += kHotThreads) {
        const int slot = q / D;
        const int32_t row = h.hot_rows[slot];
        if (want != nullptr && want[row] == 0) continue;
        const float v = acc[q];
        if (v != 0.f) rsx_atomic_add(Y + (size_t)row * D + (q - slot * D), v);
    }
}

// before: Y[hot rows] = 0 (they are summed into); after: the running layer sum S of the hot rows (S = S_init + Y or S += Y)
template <int D, bool AFTER>
__global__ __launch_bounds__(kBlock) void hot_rows_edge_kernel(rsx_spmm_hot h, const uint8_t *__restrict__ want, float *__restrict__ Y,
                                                               float *__restrict__ S, const float *__restrict__ Sinit)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int slot = t / (D / 4), k = t % (D / 4);
    if (slot >= h.num_hot) return;
    const int32_t row = h.hot_rows[slot];
    if (want != nullptr && want[row] == 0) return;                  // rsx_spmm_csr_select_rows' rule: an unwanted row is not written
    float4 *y = reinterpret_cast<float4 *>(Y + (size_t)row * D) + k;
    if constexpr (!AFTER) {
        *y = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float4 *sp = reinterpret_cast<float4 *>(S + (size_t)row * D) + k;
        float4 t4 = Sinit != nullptr ? reinterpret_cast<const float4 *>(Sinit + (size_t)row * D)[k] : *sp;
        const float4 v = *y;
        t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w;
        *sp = t4;
    }
}

unsigned grid_for(int64_t threads)
{
    int64_t blocks = (threads + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rsx_num_cus() * RSX_SPMM_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

#ifdef RSX_ABLATE
RSX_API int rsx_debug_set_graph_ablation(int mask)     // dev build only (librsx_dev.so): 1 = segment 1 of every product dropped
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_rsx_ablate), &mask, sizeof(int)) == hipSuccess ? RSX_OK : RSX_E_HIP;
}
#endif

// HOST: cut CSR rows into segments of at most max_seg non-zeros (empty rows get one empty
// segment so that Y[row] is written).  Call with out arrays NULL to get the count.
RSX_API int64_t rsx_spmm_plan(const int64_t *indptr_host, int64_t num_rows, int max_seg,
                              int32_t *seg_row_out, int64_t *seg_begin_out, int32_t *seg_len_out)
{
    if (indptr_host == nullptr || num_rows < 0 || max_seg < 1) return RSX_E_INVALID;
    int64_t n = 0;
    for (int64_t r = 0; r < num_rows; ++r) {
        const int64_t lo = indptr_host[r], hi = indptr_host[r + 1];
        int64_t p = lo;
        do {
            const int64_t len = (hi - p < max_seg) ? hi - p : max_seg;
            if (seg_row_out) { seg_row_out[n] = (int32_t)r; seg_begin_out[n] = p; seg_len_out[n] = (int32_t)len; }
            ++n;
            p += len;
        } while (p < hi);
    }
    return n;
}

static int spmm_launch(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                       int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev, const float *vals_dev,
                       const float *X, float *Y, float *S_acc, int64_t num_rows, int d, const uint8_t *nz, const uint8_t *want,
                       const float *S_init, hipStream_t st)
{
    {   // split rows add into zeros (whole rows are stored)
        const unsigned zb = (unsigned)((num_segs + kBlock - 1) / kBlock);
        switch (d) {
        case 32: hipLaunchKernelGGL(zero_split_rows_kernel<32>, dim3(zb ? zb : 1), dim3(kBlock), 0, st, seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, Y, S_acc, S_init, want); break;
        case 64: hipLaunchKernelGGL(zero_split_rows_kernel<64>, dim3(zb ? zb : 1), dim3(kBlock), 0, st, seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, Y, S_acc, S_init, want); break;
        case 128: hipLaunchKernelGGL(zero_split_rows_kernel<128>, dim3(zb ? zb : 1), dim3(kBlock), 0, st, seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, Y, S_acc, S_init, want); break;
        default: hipLaunchKernelGGL(zero_split_rows_kernel<256>, dim3(zb ? zb : 1), dim3(kBlock), 0, st, seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, Y, S_acc, S_init, want); break;
        }
    }
    const int gpw = 64 / (d / 4);
    const unsigned g = grid_for((num_segs + gpw - 1) / gpw * 64);
#define RSX_SPMM(D_) do { if (nz) hipLaunchKernelGGL((spmm_csr_kernel<D_, true>), dim3(g), dim3(kBlock), 0, st, seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, indices_dev, vals_dev, X, Y, S_acc, nz, want, S_init); \
                           else hipLaunchKernelGGL((spmm_csr_kernel<D_, false>), dim3(g), dim3(kBlock), 0, st, seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, indices_dev, vals_dev, X, Y, S_acc, nz, want, S_init); } while (0)
    switch (d) {
    case 32: RSX_SPMM(32); break;
    case 64: RSX_SPMM(64); break;
    case 128: RSX_SPMM(128); break;
    default: RSX_SPMM(256); break;
    }
#undef RSX_SPMM
    return RSX_OK;
}

RSX_API int rsx_spmm_csr(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                         int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                         const float *vals_dev, const float *X, float *Y, float *S_acc, int64_t num_rows,
                         int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(seg_row_dev && seg_begin_dev && seg_len_dev && indptr_dev && indices_dev && vals_dev && X && Y,
                  "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_rows >= 0 && num_segs >= 0, "bad shape");
    RSX_CHECK_ARG(X != Y && X != S_acc, "X must not alias an output");
    if (num_rows == 0) return RSX_OK;
    int rc = spmm_launch(seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, indices_dev, vals_dev, X, Y, S_acc, num_rows, d,
                         nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (rc != RSX_OK) return rc;
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_spmm_csr_sparse_rows(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                                     int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                                     const float *vals_dev, const float *X, const uint8_t *x_row_nonzero_dev, float *Y,
                                     float *S_acc, int64_t num_rows, int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(seg_row_dev && seg_begin_dev && seg_len_dev && indptr_dev && indices_dev && vals_dev && X && Y && x_row_nonzero_dev,
                  "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_rows >= 0 && num_segs >= 0, "bad shape");
    RSX_CHECK_ARG(X != Y && X != S_acc, "X must not alias an output");
    if (num_rows == 0) return RSX_OK;
    int rc = spmm_launch(seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, indices_dev, vals_dev, X, Y, S_acc, num_rows, d,
                         x_row_nonzero_dev, nullptr, nullptr, (hipStream_t)stream);
    if (rc != RSX_OK) return rc;
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_spmm_csr_init(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                              int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                              const float *vals_dev, const float *X, const uint8_t *x_row_nonzero_dev, const float *S_init,
                              float *Y, float *S_out, int64_t num_rows, int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(seg_row_dev && seg_begin_dev && seg_len_dev && indptr_dev && indices_dev && vals_dev && X && Y && S_init && S_out,
                  "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_rows >= 0 && num_segs >= 0, "bad shape");
    RSX_CHECK_ARG(X != Y && X != S_out && S_init != S_out && S_init != Y, "X and S_init must not alias an output");
    if (num_rows == 0) return RSX_OK;
    int rc = spmm_launch(seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, indices_dev, vals_dev, X, Y, S_out, num_rows, d,
                         x_row_nonzero_dev, nullptr, S_init, (hipStream_t)stream);
    if (rc != RSX_OK) return rc;
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_spmm_csr_select_rows(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                                     int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                                     const float *vals_dev, const float *X, const uint8_t *y_row_wanted_dev, float *Y,
                                     float *S_acc, int64_t num_rows, int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(seg_row_dev && seg_begin_dev && seg_len_dev && indptr_dev && indices_dev && vals_dev && X && Y && y_row_wanted_dev,
                  "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_rows >= 0 && num_segs >= 0, "bad shape");
    RSX_CHECK_ARG(X != Y && X != S_acc, "X must not alias an output");
    if (num_rows == 0) return RSX_OK;
    int rc = spmm_launch(seg_row_dev, seg_begin_dev, seg_len_dev, num_segs, indptr_dev, indices_dev, vals_dev, X, Y, S_acc, num_rows, d,
                         nullptr, y_row_wanted_dev, nullptr, (hipStream_t)stream);
    if (rc != RSX_OK) return rc;
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int64_t rsx_spmm_hot_capacity(int d)
{
    return rsx_dim_ok(d) ? kHotLdsFloats / d : RSX_E_INVALID;
}

RSX_API int rsx_spmm_hot_rows(const rsx_spmm_hot *hot, const float *X, const uint8_t *x_row_nonzero_dev, const uint8_t *y_row_wanted_dev,
                              const float *S_init, float *Y, float *S_acc, int64_t num_rows, int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(hot != nullptr && X && Y, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_rows > 0, "bad shape");
    RSX_CHECK_ARG(hot->num_hot >= 0 && hot->num_hot <= rsx_spmm_hot_capacity(d) && hot->num_src >= 0,
                  "num_hot must be in [0, rsx_spmm_hot_capacity(d)]");
    RSX_CHECK_ARG(X != Y && X != S_acc, "X must not alias an output");
    RSX_CHECK_ARG(S_init == nullptr || S_acc != nullptr, "S_init without S_acc");
    if (hot->num_hot == 0) return RSX_OK;
    RSX_CHECK_ARG(hot->hot_rows && (hot->num_src == 0 || (hot->src_rows && hot->src_ptr && hot->src_slot && hot->src_val)), "null plan array");
    hipStream_t st = (hipStream_t)stream;
    const unsigned eb = (unsigned)ceil_div64((int64_t)hot->num_hot * (d / 4), kBlock);
    int64_t wgs = rsx_num_cus();
    if (wgs > hot->num_src) wgs = hot->num_src;
#define RSX_HOT(D_) do { \
        hipLaunchKernelGGL((hot_rows_edge_kernel<D_, false>), dim3(eb), dim3(kBlock), 0, st, *hot, y_row_wanted_dev, Y, S_acc, S_init); \
        if (wgs > 0) hipLaunchKernelGGL(spmm_hot_rows_kernel<D_>, dim3((unsigned)wgs), dim3(kHotThreads), 0, st, *hot, X, x_row_nonzero_dev, y_row_wanted_dev, Y); \
        if (S_acc != nullptr) hipLaunchKernelGGL((hot_rows_edge_kernel<D_, true>), dim3(eb), dim3(kBlock), 0, st, *hot, y_row_wanted_dev, Y, S_acc, S_init); \
    } while (0)
    switch (d) {
    case 32: RSX_HOT(32); break;
    case 64: RSX_HOT(64); break;
    case 128: RSX_HOT(128); break;
    default: RSX_HOT(256); break;
    }
#undef RSX_HOT
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

__global__ __launch_bounds__(256) void mark_batch_rows_kernel(uint8_t *__restrict__ flags, const int32_t *__restrict__ u,
                                                              const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                              int64_t B, int64_t item_offset)
{
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
        const int32_t ib = i[b];
        if (ib < 0) continue;
        flags[u[b]] = 1;                       // (benign same-value races between triplets that share a row)
        flags[item_offset + ib] = 1;
        flags[item_offset + j[b]] = 1;
    }
}

RSX_API int rsx_spmm_mark_batch_rows(uint8_t *flags_dev, int64_t num_rows, const int32_t *u_dev, const int32_t *i_dev,
                                     const int32_t *j_dev, int64_t batch, int64_t item_offset, rsx_stream_t stream)
{
    RSX_CHECK_ARG(flags_dev != nullptr && num_rows > 0 && batch >= 0 && item_offset >= 0 && item_offset <= num_rows, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(flags_dev, 0, (size_t)num_rows, st) != hipSuccess) {
        rsx_set_error("rsx_spmm_mark_batch_rows: memset failed");
        return RSX_E_HIP;
    }
    if (batch == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev && j_dev, "null index pointer");
    int64_t blocks = (batch + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mark_batch_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flags_dev, u_dev, i_dev, j_dev, batch, item_offset);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

template <int D4>
__global__ __launch_bounds__(256) void scale_flagged_rows_kernel(float4 *__restrict__ X, const uint8_t *__restrict__ flags, int64_t num_rows,
                                                                 float alpha)
{
    // one thread per quad of a row: a flagged row's D / 4 quads are handled by D / 4 consecutive threads; alpha == 0 stores zeros
    // (whatever the row held)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / D4;
    if (row >= num_rows || flags[row] == 0) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (alpha != 0.0f) { v = X[t]; v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }
    X[t] = v;
}

RSX_API int rsx_spmm_scale_rows(float *X, const uint8_t *flags_dev, int64_t num_rows, int d, float alpha, rsx_stream_t stream)
{
    RSX_CHECK_ARG(X && flags_dev && num_rows >= 0 && rsx_dim_ok(d), "bad arguments");
    if (num_rows == 0) return RSX_OK;
    const int64_t threads = num_rows * (d / 4);
    const unsigned g = (unsigned)((threads + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
    case 32: hipLaunchKernelGGL(scale_flagged_rows_kernel<8>, dim3(g), dim3(256), 0, st, (float4 *)X, flags_dev, num_rows, alpha); break;
    case 64: hipLaunchKernelGGL(scale_flagged_rows_kernel<16>, dim3(g), dim3(256), 0, st, (float4 *)X, flags_dev, num_rows, alpha); break;
    case 128: hipLaunchKernelGGL(scale_flagged_rows_kernel<32>, dim3(g), dim3(256), 0, st, (float4 *)X, flags_dev, num_rows, alpha); break;
    default: hipLaunchKernelGGL(scale_flagged_rows_kernel<64>, dim3(g), dim3(256), 0, st, (float4 *)X, flags_dev, num_rows, alpha); break;
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_scale(float *X, int64_t n, float alpha, rsx_stream_t stream)
{
    RSX_CHECK_ARG(X != nullptr && n >= 0 && n % 4 == 0, "n must be a multiple of 4");
    if (n == 0) return RSX_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, (float4 *)X, n / 4, alpha);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}
