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
// (5 GB of the product's 10.6 GB at the configs[4] shape; 0.68 ms of its 2.33 ms).  Turned round -- walk the SOURCE rows in order, read
// each once, add a x into the accumulators of the hot rows it belongs to -- the same entries cost one streaming pass over the source rows
// that have a hot neighbour (0.5 GB).
//   * the accumulators live in REGISTERS: each of a workgroup's 16 wavefronts owns H / 16 hot slots (16 registers per lane), and the
//     caller's plan hands every wavefront ITS entries per chunk of source rows.  (First form: accumulators in LDS, ds_add_f32 from every
//     lane group: 4.3 ms -- LDS float atomics retire about half a lane per clock and CU; profiles/r06_exp_spmm_hot_rows.txt.)
//   * the source rows of a chunk are staged through LDS once per workgroup (the only HBM read of them) and read from there per entry;
//   * a very long row is given several slots by the plan (its entries dealt round), so that no wavefront carries it alone; at the end
//     every wavefront adds its rows to Y (pre-zeroed) with global atomics -- slots x workgroups row updates, not one per entry.
// The caller's segment plan (rsx_spmm_plan) simply owns no segment for the hot rows; rsx_spmm_hot_rows computes them.
constexpr int kHotThreads = 1024, kHotWaves = kHotThreads / 64;
constexpr int kHotSlotFloats = 16384;         // H * D: 16 accumulator registers per lane in each of the 16 wavefronts
constexpr int kHotTileFloats = 8192;          // LDS tile of source rows: chunk_rows * D <= 8192 (32 KB)

template <int D>
__global__ __launch_bounds__(kHotThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmm_hot_rows_kernel(rsx_spmm_hot h, const float *__restrict__ X, const uint8_t *__restrict__ nz,
                                                                    const uint8_t *__restrict__ want, float *__restrict__ Y)
{
    __shared__ __attribute__((aligned(16))) float tile[kHotTileFloats];
    constexpr int R = kHotSlotFloats / D / kHotWaves;               // hot slots per wavefront
    constexpr int NX = D >= 64 ? D / 64 : 1;                        // registers a source row takes per lane
    constexpr int NACC = R * D / 64;                                // = 16
    static_assert(NACC == 16, "16 accumulator registers per lane");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = h.chunk_rows;
    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
    const int64_t nchunks = ceil_div64(h.num_src, K);
    // TWO chunks ahead.  A chunk's loads are two dependent levels (row id -> row of X; pointer -> entries): the ids and pointers of chunk
    // n + 2 and the rows and entries of chunk n + 1 travel while chunk n is added, each level a whole iteration old when it is needed --
    // per chunk the loop itself only stores the tile, meets at two barriers and computes.
    static_assert(kHotTileFloats / 4 == 2 * kHotThreads, "two float4 of the tile per thread");
    int32_t ia[2] = {-1, -1};                                       // stage A: my two rows' ids (-1: nothing / flagged off) ...
    int64_t pa0 = 0, pa1 = 0;                                       // ... and this wavefront's entry range
    float4 tr[2];                                                   // stage B: my two float4 of the tile ...
    uint32_t pcode = 0u; float pval = 0.f; int64_t pe0 = 0, pe1 = 0;   // ... and the wavefront's first 64 entries
    auto fetch_ids = [&](int64_t ch) __attribute__((always_inline)) {
        ia[0] = ia[1] = -1; pa0 = pa1 = 0;
        if (ch >= nchunks) return;
        const int rows = (int)((h.num_src - ch * K < K) ? h.num_src - ch * K : K);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = tid + t * kHotThreads;
            if (q < rows * (D / 4)) {
                const int32_t c = h.src_rows[ch * K + q / (D / 4)];
                ia[t] = (nz != nullptr && nz[c] == 0) ? -1 : c;     // (a row whose flag is off is staged as zeros)
            }
        }
        pa0 = h.cw_ptr[ch * kHotWaves + wave]; pa1 = h.cw_ptr[ch * kHotWaves + wave + 1];
    };
    auto fetch_data = [&]() __attribute__((always_inline)) {       // from stage A's registers
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c4 = (tid + t * kHotThreads) % (D / 4);
            const float *src = ia[t] < 0 ? g_zero_row : X + (size_t)ia[t] * D;
            tr[t] = reinterpret_cast<const float4 *>(src)[c4];
        }
        pe0 = pa0; pe1 = pa1; pcode = 0u; pval = 0.f;
        if (pe0 + lane < pe1) { pcode = h.ent_code[pe0 + lane]; pval = h.ent_val[pe0 + lane]; }
    };
    fetch_ids(blockIdx.x);
    fetch_data();
    fetch_ids((int64_t)blockIdx.x + gridDim.x);
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        reinterpret_cast<float4 *>(tile)[tid] = tr[0];
        reinterpret_cast<float4 *>(tile)[tid + kHotThreads] = tr[1];
        const int64_t e0 = pe0, e1 = pe1;
        uint32_t codev = pcode; float valv = pval;
        __syncthreads();
        fetch_data();                                               // chunk ch + G (its ids arrived an iteration ago)
        fetch_ids(ch + 2 * (int64_t)gridDim.x);
        // the wavefront's entries of this chunk are SORTED BY SLOT, rw_off[0 .. R] cutting them (relative to e0): the slot of an entry is a
        // compile-time constant inside each of the R unrolled sections below, so its accumulator registers are addressed statically.  (A
        // jump on the slot per entry made the compiler carry all 16 accumulators through every branch: ~50 register moves per entry,
        // 533 us; profiles/r06_exp_spmm_hot_rows.txt)
        const uint32_t offv = (lane <= R) ? (uint32_t)h.rw_off[(ch * kHotWaves + wave) * (R + 1) + lane] : 0u;
        for (int64_t base = e0; base < e1; base += 64) {
            const int n = __builtin_amdgcn_readfirstlane((int)((e1 - base < 64) ? e1 - base : 64));
            if (base > e0) {                                        // (more than 64 entries for this wavefront in one chunk: rare)
                codev = 0u; valv = 0.f;
                if (lane < n) { codev = h.ent_code[base + lane]; valv = h.ent_val[base + lane]; }
            }
            const int w0 = __builtin_amdgcn_readfirstlane((int)(base - e0));   // this window holds the entries [w0, w0 + n) of the chunk
            // (scalar loop control and two entries per trip: the first form of this loop spent 10 vector instructions per entry -- lane reads,
            //  the loop counter in a vector register, address arithmetic -- on one useful multiply-add: 92M wave instructions per launch)
#pragma unroll
            for (int r_ = 0; r_ < R; ++r_) {
                int q0 = (int)__builtin_amdgcn_readlane((int)offv, r_) - w0, q1 = (int)__builtin_amdgcn_readlane((int)offv, r_ + 1) - w0;
                q0 = q0 < 0 ? 0 : q0; q1 = q1 > n ? n : q1;
                q0 = __builtin_amdgcn_readfirstlane(q0); q1 = __builtin_amdgcn_readfirstlane(q1);     // (wave-uniform: say so)
                auto entry_off = [&](int q) __attribute__((always_inline)) { return (int)((uint32_t)__builtin_amdgcn_readlane((int)codev, q) & 0xFFu) * D; };
                auto entry_val = [&](int q) __attribute__((always_inline)) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, valv), q)); };
                auto load_row = [&](int off, float (&x)[NX]) __attribute__((always_inline)) {
                    if constexpr (D >= 64) {
#pragma unroll
                        for (int m = 0; m < NX; ++m) x[m] = tile[off + m * 64 + lane];
                    } else {
                        x[0] = tile[off + (lane & (D - 1))];
                    }
                };
                auto add_row = [&](float a, const float (&x)[NX]) __attribute__((always_inline)) {
                    if constexpr (D >= 64) {
#pragma unroll
                        for (int m = 0; m < NX; ++m) acc[r_ * NX + m] = fmaf(a, x[m], acc[r_ * NX + m]);
                    } else {
                        acc[r_ / 2] = fmaf(((lane >> 5) == (r_ & 1)) ? a : 0.f, x[0], acc[r_ / 2]);
                    }
                };
                int q = q0;
                for (; q + 2 <= q1; q += 2) {                       // both rows requested from LDS before either is used
                    float xa[NX], xb[NX];
                    load_row(entry_off(q), xa);
                    load_row(entry_off(q + 1), xb);
                    add_row(entry_val(q), xa);
                    add_row(entry_val(q + 1), xb);
                }
                if (q < q1) { float xa[NX]; load_row(entry_off(q), xa); add_row(entry_val(q), xa); }
            }
        }
        __syncthreads();                                            // (the tile is overwritten by the next chunk)
    }
    // the wavefront's slots -> Y: register j of lane l holds element (j * 64 + l) of its [R x D] block
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const int f = j * 64 + lane, r = f / D, el = f % D;
        const int32_t row = h.hot_rows[wave * R + r];
        if (row < 0 || (want != nullptr && want[row] == 0)) continue;
        if (acc[j] != 0.f) rsx_atomic_add(Y + (size_t)row * D + el, acc[j]);
    }
}

// before: Y[hot rows] = 0 (they are summed into); after: the running layer sum S of the hot rows (S = S_init + Y or S += Y).  Over the
// DISTINCT hot rows (a row with several slots appears once in uniq_rows)
template <int D, bool AFTER>
__global__ __launch_bounds__(kBlock) void hot_rows_edge_kernel(rsx_spmm_hot h, const uint8_t *__restrict__ want, float *__restrict__ Y,
                                                               float *__restrict__ S, const float *__restrict__ Sinit)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int slot = t / (D / 4), k = t % (D / 4);
    if (slot >= h.num_uniq) return;
    const int32_t row = h.uniq_rows[slot];
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
    return rsx_dim_ok(d) ? kHotSlotFloats / d : RSX_E_INVALID;
}

RSX_API int64_t rsx_spmm_hot_chunk_rows(int d)
{
    if (!rsx_dim_ok(d)) return RSX_E_INVALID;
    const int k = kHotTileFloats / d;
    return k < 64 ? k : 64;
}

RSX_API int rsx_spmm_hot_rows(const rsx_spmm_hot *hot, const float *X, const uint8_t *x_row_nonzero_dev, const uint8_t *y_row_wanted_dev,
                              const float *S_init, float *Y, float *S_acc, int64_t num_rows, int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(hot != nullptr && X && Y, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_rows > 0, "bad shape");
    RSX_CHECK_ARG(hot->num_slots == rsx_spmm_hot_capacity(d) && hot->chunk_rows == rsx_spmm_hot_chunk_rows(d),
                  "the plan was built for another row width: num_slots = rsx_spmm_hot_capacity(d), chunk_rows = rsx_spmm_hot_chunk_rows(d)");
    RSX_CHECK_ARG(hot->num_src >= 0 && hot->num_uniq >= 0 && hot->num_uniq <= hot->num_slots, "bad plan sizes");
    RSX_CHECK_ARG(X != Y && X != S_acc, "X must not alias an output");
    RSX_CHECK_ARG(S_init == nullptr || S_acc != nullptr, "S_init without S_acc");
    if (hot->num_uniq == 0) return RSX_OK;
    RSX_CHECK_ARG(hot->hot_rows && hot->uniq_rows && (hot->num_src == 0 || (hot->src_rows && hot->cw_ptr && hot->rw_off && hot->ent_code && hot->ent_val)),
                  "null plan array");
    hipStream_t st = (hipStream_t)stream;
    const unsigned eb = (unsigned)ceil_div64((int64_t)hot->num_uniq * (d / 4), kBlock);
    const int64_t nchunks = ceil_div64(hot->num_src, hot->chunk_rows);
    int64_t wgs = 2 * (int64_t)rsx_num_cus();           // (1024 threads, 32 KB of LDS: two workgroups fill a CU's wave slots)
    if (wgs > nchunks) wgs = nchunks;
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
