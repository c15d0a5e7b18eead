// tools/mfma_peak.hip : what v_mfma_f32_32x32x2_f32 delivers with NOTHING around it -- no memory, no LDS, no barriers: every wavefront
// issues chains of independent MFMAs on register operands.  The ceiling of the scoring product (csrc/rsx_score.hip) on this part, next to
// the nominal 157.3 TFLOP/s (256 CUs x 4 SIMDs x 256 flop / clk x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACCS>
__global__ __launch_bounds__(256) void mfma_chain(float *out, int iters, float a, float b)
{
    f32x16 acc[ACCS];
#pragma unroll
    for (int q = 0; q < ACCS; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < ACCS; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < ACCS; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (s == 12345.678f) out[threadIdx.x] = s;      // (never true: keeps the chain alive)
}

// the same chain on operands that CHANGE every step and differ from lane to lane (pseudo-random mantissas, like real embeddings): the
// data path toggles, the power per MFMA is what a real product draws
template <int ACCS>
__global__ __launch_bounds__(256) void mfma_chain_data(float *out, int iters, const float *seed)
{
    f32x16 acc[ACCS];
#pragma unroll
    for (int q = 0; q < ACCS; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = seed[(threadIdx.x * 8 + k) & 1023]; b[k] = seed[(threadIdx.x * 8 + 4 + k) & 1023]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int q = 0; q < ACCS; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + q) & 3], b[(u + 2 * q) & 3], acc[q], 0, 0, 0);
            // (one cheap VALU op per step keeps the operands moving: sign flips and a mantissa rotation, magnitudes stay ~0.1)
            a[u & 3] = __uint_as_float((__float_as_uint(a[u & 3]) ^ 0x80155555u) | 0x3D000000u);
            b[u & 3] = __uint_as_float((__float_as_uint(b[u & 3]) ^ 0x802AAAAAu) | 0x3D000000u);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < ACCS; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// the inner loop of the scoring product (csrc/rsx_score.hip) and nothing else: per step four fragment reads from LDS (one step ahead) and four
// MFMAs into a 2 x 2 block of accumulators, 16 steps per "chunk" -- no global memory, no DMA, no barriers, no epilogue
__global__ __launch_bounds__(256, 4) void mfma_lds_loop(float *out, int chunks, int random_data)
{
    __shared__ float As[128 * 32 + 16], Bs[128 * 32 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, hi = lane >> 5, l31 = lane & 31;
    for (int q = tid; q < 128 * 32 + 16; q += 256) {
        As[q] = 0.001f * (float)((q * 37) % 201 - 100); Bs[q] = 0.001f * (float)((q * 53) % 199 - 99);
        if (random_data) {       // full-entropy mantissas and random signs, magnitudes ~0.1 like N(0, 0.1^2) embeddings
            unsigned x = (unsigned)q * 2654435761u + blockIdx.x * 40503u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            unsigned y = x * 3266489917u; y ^= y >> 16;
            As[q] = __uint_as_float((x & 0x807FFFFFu) | 0x3D800000u);      // +-[0.0625, 0.125)
            Bs[q] = __uint_as_float((y & 0x807FFFFFu) | 0x3D800000u);
        }
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int xs = (l31 & 7) << 2;
    const float *ap = As + (wr * 64 + l31) * 32 + ((wr * 64 + l31) >> 3) + hi;
    const float *bp = Bs + (wc * 64 + l31) * 32 + ((wc * 64 + l31) >> 3) + hi;
    for (int c = 0; c < chunks; ++c) {
        float a0 = ap[xs], a1 = ap[32 * 32 + 4 + xs], b0 = bp[xs], b1 = bp[32 * 32 + 4 + xs];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < 32) {
                const int o = (kk + 2) ^ xs;
                na0 = ap[o]; na1 = ap[32 * 32 + 4 + o]; nb0 = bp[o]; nb1 = bp[32 * 32 + 4 + o];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        asm volatile("" ::: "memory");      // (the next chunk reads LDS again)
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 12345.678f) out[tid] = s;
}

// ... the same loop WITH the kernel's global -> LDS traffic: per chunk every wavefront issues its eight global_load_lds_dwordx4 (8 rows x 128 B
// each, rows of a `rows`-row fp32 table picked like the kernel picks them), then the two barriers -- the scoring kernel minus prologue and epilogue
template <int MODE>
__device__ __forceinline__ void mfma_dma_body(float *out, int chunks, const float *table, int rows, int d)
{
    constexpr int NB = MODE == 3 ? 2 : 1;
    constexpr int TS = 128 * 32 + 16;
    __shared__ __attribute__((aligned(16))) float As[NB * TS], Bs[NB * TS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, hi = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int xs = (l31 & 7) << 2;
    const float *ap = As + (wr * 64 + l31) * 32 + ((wr * 64 + l31) >> 3) + hi;
    const float *bp = Bs + (wc * 64 + l31) * 32 + ((wc * 64 + l31) >> 3) + hi;
    const int kq = tid & 7, srow = tid >> 3, gq = kq ^ (srow & 7);
    unsigned base = blockIdx.x * 2654435761u;
    auto dma = [&](int c, int buf) __attribute__((always_inline)) {
        const int k0 = (c * 32) % d;
        if (k0 == 0) base = base * 1664525u + 1013904223u;      // a new pair of 128-row panels every d / 32 chunks
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned ra = (base + (unsigned)(srow + 32 * n)) % (unsigned)rows, rb = (base * 7u + (unsigned)(srow + 32 * n)) % (unsigned)rows;
            __builtin_amdgcn_global_load_lds(table + (size_t)ra * d + k0 + 4 * gq, As + buf * TS + (32 * n + 8 * wid) * 32 + (4 * n + wid), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(table + (size_t)rb * d + k0 + 4 * gq, Bs + buf * TS + (32 * n + 8 * wid) * 32 + (4 * n + wid), 16, 0, 0);
        }
    };
    if (MODE == 1 || MODE == 3) { dma(0, 0); }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    for (int c = 0; c < chunks; ++c) {
        const int buf = MODE == 3 ? (c & 1) : 0;
        if (MODE == 0 || MODE == 2) dma(c, 0);
        if (MODE == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (c + 1 < chunks) dma(c + 1, buf ^ 1); }
        else if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __syncthreads();
        const float *apc = ap + buf * TS, *bpc = bp + buf * TS;
        float a0 = apc[xs], a1 = apc[32 * 32 + 4 + xs], b0 = bpc[xs], b1 = bpc[32 * 32 + 4 + xs];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < 32) {
                const int o = (kk + 2) ^ xs;
                na0 = apc[o]; na1 = apc[32 * 32 + 4 + o]; nb0 = bpc[o]; nb1 = bpc[32 * 32 + 4 + o];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (MODE == 0 || MODE == 1) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 12345.678f) out[tid] = s;
}
// (one plain kernel per mode: a templated __global__ with this body loses its host stub at link time -- the same compiler quirk
//  csrc/rsx_score.hip works around)
__global__ __launch_bounds__(256, 4) void mfma_dma_loop0(float *o, int c, const float *t, int r, int d) { mfma_dma_body<0>(o, c, t, r, d); }
__global__ __launch_bounds__(256, 4) void mfma_dma_loop1(float *o, int c, const float *t, int r, int d) { mfma_dma_body<1>(o, c, t, r, d); }
__global__ __launch_bounds__(256, 4) void mfma_dma_loop2(float *o, int c, const float *t, int r, int d) { mfma_dma_body<2>(o, c, t, r, d); }
__global__ __launch_bounds__(256, 4) void mfma_dma_loop3(float *o, int c, const float *t, int r, int d) { mfma_dma_body<3>(o, c, t, r, d); }


// the same structure for other workgroup SHAPES: WR x WC wavefronts of 64 x 64 outputs each (tile 64 WR x 64 WC), every wavefront issuing its share
// of the (64 WR + 64 WC) / 8 LDS-DMA instructions of a chunk; DB: two LDS buffers, the next chunk's DMA in flight, one barrier per chunk.
// The DMA instruction count per MFMA falls with the tile: 2 x 2: 8 per wavefront and chunk, 4 x 2: 6, 4 x 4: 4.
template <int WR, int WC, bool DB>
__device__ __forceinline__ void shape_body(float *out, int chunks, const float *table, int rows, int d)
{
    constexpr int BMs = 64 * WR, BNs = 64 * WC, NW = WR * WC, NB = DB ? 2 : 1;
    constexpr int TA = BMs * 32 + BMs / 8, TB = BNs * 32 + BNs / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds, *Bs = lds + NB * TA;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid / WC, wc = wid % WC, hi = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int xs = (l31 & 7) << 2;
    const float *ap = As + (wr * 64 + l31) * 32 + ((wr * 64 + l31) >> 3) + hi;
    const float *bp = Bs + (wc * 64 + l31) * 32 + ((wc * 64 + l31) >> 3) + hi;
    const int kq = lane & 7, r8 = lane >> 3, gq = kq ^ r8;      // a DMA instruction: 8 rows x 8 quads; row & 7 = r8
    unsigned base = blockIdx.x * 2654435761u;
    constexpr int PIECES = (BMs + BNs) / 8, PER = PIECES / NW;      // 8-row pieces of both tiles, dealt to the wavefronts
    static_assert(PIECES % NW == 0, "pieces must divide");
    auto dma = [&](int c, int buf) __attribute__((always_inline)) {
        const int k0 = (c * 32) % d;
        if (k0 == 0) base = base * 1664525u + 1013904223u;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int piece = wid * PER + q;                         // < BMs / 8: a piece of A, else of B
            const bool isA = piece < BMs / 8;
            const int prow = (isA ? piece : piece - BMs / 8) * 8;
            const unsigned gr = ((isA ? base : base * 7u) + (unsigned)(prow + r8)) % (unsigned)rows;
            float *dst = (isA ? As + buf * TA : Bs + buf * TB) + prow * 32 + (prow >> 3);
            __builtin_amdgcn_global_load_lds(table + (size_t)gr * d + k0 + 4 * gq, dst, 16, 0, 0);
        }
    };
    if (DB) dma(0, 0);
    for (int c = 0; c < chunks; ++c) {
        const int buf = DB ? (c & 1) : 0;
        if (!DB) dma(c, 0);
        if (DB) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (c + 1 < chunks) dma(c + 1, buf ^ 1); }
        else __syncthreads();
        const float *apc = ap + buf * TA, *bpc = bp + buf * TB;
        float a0 = apc[xs], a1 = apc[32 * 32 + 4 + xs], b0 = bpc[xs], b1 = bpc[32 * 32 + 4 + xs];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < 32) {
                const int o = (kk + 2) ^ xs;
                na0 = apc[o]; na1 = apc[32 * 32 + 4 + o]; nb0 = bpc[o]; nb1 = bpc[32 * 32 + 4 + o];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (!DB) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 12345.678f) out[tid] = s;
}
__global__ __launch_bounds__(256, 4) void shape_2x2(float *o, int c, const float *t, int r, int d) { shape_body<2, 2, false>(o, c, t, r, d); }
__global__ __launch_bounds__(512, 2) void shape_4x2(float *o, int c, const float *t, int r, int d) { shape_body<4, 2, false>(o, c, t, r, d); }
__global__ __launch_bounds__(512, 2) void shape_4x2db(float *o, int c, const float *t, int r, int d) { shape_body<4, 2, true>(o, c, t, r, d); }
__global__ __launch_bounds__(1024, 1) void shape_4x4(float *o, int c, const float *t, int r, int d) { shape_body<4, 4, false>(o, c, t, r, d); }
__global__ __launch_bounds__(1024, 1) void shape_4x4db(float *o, int c, const float *t, int r, int d) { shape_body<4, 4, true>(o, c, t, r, d); }

// Round 6 (the review's item 7): "a feed that is not LDS-DMA pieces of 1 KiB for BOTH operands" -- the A operand (the user tile: it is the same
// for every item tile of a pass and lives in L2) comes by DIRECT global loads in MFMA layout, only B through the LDS-DMA: half the DMA pieces per
// MFMA (4 per wavefront and chunk instead of 8).  A lane holds its two rows' 16 contraction values of the chunk: lane (l31, hi) takes k = 16 hi ..
// 16 hi + 15 (four dwordx4 per row block = 32 VGPRs per chunk; B's fragment reads follow the same k mapping), and the NEXT chunk's A travels while
// this chunk's MFMAs run (DOUBLE: 64 VGPRs of A) or is fetched at the top of the chunk (SINGLE: 32).
template <bool DOUBLE>
__device__ __forceinline__ void mfma_adirect_body(float *out, int chunks, const float *table, int rows, int d)
{
    constexpr int TS = 128 * 32 + 16;
    __shared__ __attribute__((aligned(16))) float Bs[TS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, hi = lane >> 5, l31 = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const int xs = (l31 & 7) << 2, hb = (16 * hi) ^ xs;
    const float *bp = Bs + (wc * 64 + l31) * 32 + ((wc * 64 + l31) >> 3);
    const int kq = tid & 7, srow = tid >> 3, gq = kq ^ (srow & 7);
    unsigned base = blockIdx.x * 2654435761u;
    unsigned abase = base;
    auto dma_b = [&](int c) __attribute__((always_inline)) {
        const int k0 = (c * 32) % d;
        if (k0 == 0) base = base * 1664525u + 1013904223u;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned rb = (base * 7u + (unsigned)(srow + 32 * n)) % (unsigned)rows;
            __builtin_amdgcn_global_load_lds(table + (size_t)rb * d + k0 + 4 * gq, Bs + (32 * n + 8 * wid) * 32 + (4 * n + wid), 16, 0, 0);
        }
    };
    auto load_a = [&](int c, float4 (&a)[2][4]) __attribute__((always_inline)) {
        const int k0 = (c * 32) % d;
        if (k0 == 0) abase = abase * 1664525u + 1013904223u;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const unsigned ra = (abase + (unsigned)(wr * 64 + 32 * blk + l31)) % (unsigned)rows;
            const float4 *src = reinterpret_cast<const float4 *>(table + (size_t)ra * d + k0 + 16 * hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[blk][q] = src[q];
        }
    };
    float4 acur[2][4], anext[2][4];
    if (DOUBLE) load_a(0, anext);
    for (int c = 0; c < chunks; ++c) {
        dma_b(c);
        if (DOUBLE) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) acur[blk][q] = anext[blk][q];
        } else {
            load_a(c, acur);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (DOUBLE && c + 1 < chunks) load_a(c + 1, anext);
        float b0 = bp[hb], b1 = bp[32 * 32 + 4 + hb];
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            float nb0 = 0.f, nb1 = 0.f;
            if (s_ + 1 < 16) {
                const int o = (hb ^ ((s_ + 1) & ~3)) | ((s_ + 1) & 3);
                nb0 = bp[o]; nb1 = bp[32 * 32 + 4 + o];
            }
            const float a0 = reinterpret_cast<const float *>(&acur[0][s_ >> 2])[s_ & 3], a1 = reinterpret_cast<const float *>(&acur[1][s_ >> 2])[s_ & 3];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            b0 = nb0; b1 = nb1;
        }
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (s == 12345.678f) out[tid] = s;
}
__global__ __launch_bounds__(256, 4) void mfma_adirect_single(float *o, int c, const float *t, int r, int d) { mfma_adirect_body<false>(o, c, t, r, d); }
__global__ __launch_bounds__(256, 3) void mfma_adirect_double3(float *o, int c, const float *t, int r, int d) { mfma_adirect_body<true>(o, c, t, r, d); }
__global__ __launch_bounds__(256, 4) void mfma_adirect_double4(float *o, int c, const float *t, int r, int d) { mfma_adirect_body<true>(o, c, t, r, d); }

template <int ACCS>
void run(int waves_per_simd, int cus, int iters = 2000)
{
    float *out;
    (void)hipMalloc(&out, 4096);
    dim3 grid(cus * waves_per_simd), block(256);      // a 256-thread workgroup = one wavefront per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_chain<ACCS><<<grid, block>>>(out, 10, 1.0f, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        mfma_chain<ACCS><<<grid, block>>>(out, iters, 1.0f, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)grid.x * 4 /*waves*/ * iters * 16.0 * ACCS * 4096.0;      // 32 x 32 x 2 x 2 flop per MFMA
    printf("%d independent accumulators, %d wavefronts per SIMD: %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", ACCS, waves_per_simd, best,
           flop / best / 1e9, flop / best / 1e9 / 157.3);
    hipFree(out);
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    if (argc > 1 && argv[1][0] == 'a') {        // `tools/mfma_peak adirect`: only round 6's A-by-direct-loads feed against the kernel's own structure
        float *out;
        (void)hipMalloc(&out, 4096);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rows : {24576, 1000000}) {
            float *table;
            (void)hipMalloc(&table, (size_t)rows * 128 * 4);
            (void)hipMemset(table, 0x3d, (size_t)rows * 128 * 4);
            struct { const char *name; void (*k)(float *, int, const float *, int, int); int per_cu; } ks[] = {
                {"LDS-DMA for A and B + two barriers per chunk (the kernel's structure), 4 workgroups per CU", mfma_dma_loop0, 4},
                {"A by direct loads at the top of the chunk (32 VGPRs), B by LDS-DMA, 4 per CU", mfma_adirect_single, 4},
                {"A by direct loads one chunk ahead (64 VGPRs), B by LDS-DMA, 3 per CU", mfma_adirect_double3, 3},
                {"A by direct loads one chunk ahead, compiled for 4 per CU", mfma_adirect_double4, 4}};
            for (auto &kk : ks) {
                dim3 grid(p.multiProcessorCount * kk.per_cu), block(256);
                const int chunks = 500;
                const double flop = (double)grid.x * 4 * chunks * 16.0 * 4 * 4096.0;
                printf("table of %d rows x 128, %s:", rows, kk.name);
                for (int rep = 0; rep < 8; ++rep) {
                    (void)hipEventRecord(e0);
                    kk.k<<<grid, block>>>(out, chunks, table, rows, 128);
                    (void)hipEventRecord(e1);
                    if (rep % 2 == 1) {
                        (void)hipEventSynchronize(e1);
                        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                        printf(" %.1f", flop / ms / 1e9);
                    }
                }
                printf(" TFLOP/s  [%s]\n", hipGetErrorString(hipGetLastError()));
            }
            (void)hipFree(table);
        }
        return 0;
    }
    for (int w : {1, 2, 4}) { run<1>(w, p.multiProcessorCount); run<2>(w, p.multiProcessorCount); run<4>(w, p.multiProcessorCount); }
    // the shape of the scoring product is 4 accumulators x 4 wavefronts per SIMD: which of the two makes 0.80 of it?  (launches of equal length)
    printf("--- equal work per launch (~3.4 ms at full rate)\n");
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run<1>(w, p.multiProcessorCount, 8000 / w); run<2>(w, p.multiProcessorCount, 4000 / w); run<3>(w, p.multiProcessorCount, 2667 / w);
        run<4>(w, p.multiProcessorCount, 2000 / w);
    }
    // SUSTAINED: back-to-back launches of ~3.4 ms each for ~0.3 s, the rate of every tenth one -- does the part hold its clock under fp32 MFMA?
    {
        float *out;
        (void)hipMalloc(&out, 4096);
        dim3 grid(p.multiProcessorCount * 4), block(256);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2000;
        const double flop = (double)grid.x * 4 * iters * 16.0 * 1 * 4096.0;
        printf("sustained (1 accumulator, 4 wavefronts per SIMD, launches of ~3.4 ms back to back):");
        for (int rep = 0; rep < 100; ++rep) {
            (void)hipEventRecord(e0);
            mfma_chain<1><<<grid, block>>>(out, iters, 1.0f, 1.0f);
            (void)hipEventRecord(e1);
            if (rep % 10 == 9) {
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                printf(" %.1f", flop / ms / 1e9);
            }
        }
        printf(" TFLOP/s\n");
        // ... and with operands that toggle like real data
        float h[1024];
        unsigned x = 12345u;
        for (int k = 0; k < 1024; ++k) { x = x * 1664525u + 1013904223u; h[k] = ((int)(x >> 8) % 2001 - 1000) * 1e-4f; }
        float *seed;
        (void)hipMalloc(&seed, sizeof(h));
        (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
        for (int accs = 2; accs <= 3; ++accs) {
            printf("sustained, operands changing every step (%d accumulators, 4 wavefronts per SIMD, ~3.4 ms launches back to back):", accs);
            const double fl = (double)grid.x * 4 * (accs == 2 ? 1000 : 667) * 16.0 * accs * 4096.0;
            for (int rep = 0; rep < 100; ++rep) {
                (void)hipEventRecord(e0);
                if (accs == 2) mfma_chain_data<2><<<grid, block>>>(out, 1000, seed); else mfma_chain_data<3><<<grid, block>>>(out, 667, seed);
                (void)hipEventRecord(e1);
                if (rep % 10 == 9) {
                    (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    printf(" %.1f", fl / ms / 1e9);
                }
            }
            printf(" TFLOP/s\n");
        }
    }
    // the scoring product's inner loop alone, four workgroups per CU like the real kernel
    {
        float *out;
        (void)hipMalloc(&out, 4096);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int random_data : {0, 1})
        for (int wg_per_cu : {1, 2, 4}) {
            dim3 grid(p.multiProcessorCount * wg_per_cu), block(256);
            const int chunks = 2000 / wg_per_cu;
            const double flop = (double)grid.x * 4 * chunks * 16.0 * 4 * 4096.0;
            printf("scoring inner loop only (LDS fragment reads + 2 x 2 MFMAs, %d workgroups per CU, %s):", wg_per_cu,
                   random_data ? "RANDOM full-entropy operands" : "201 distinct operand values");
            for (int rep = 0; rep < 60; ++rep) {
                (void)hipEventRecord(e0);
                mfma_lds_loop<<<grid, block>>>(out, chunks, random_data);
                (void)hipEventRecord(e1);
                if (rep % 10 == 9) {
                    (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    printf(" %.1f", flop / ms / 1e9);
                }
            }
            printf(" TFLOP/s\n");
        }
    }
    // ... and in workgroups as SHORT as the real kernel's: one 128 x 128 x 128 tile each (4 chunks), ~50 000 of them per launch
    {
        float *out;
        (void)hipMalloc(&out, 4096);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int chunks : {4, 8, 16, 64}) {
            dim3 grid(50048 * 4 / chunks), block(256);
            const double flop = (double)grid.x * 4 * chunks * 16.0 * 4 * 4096.0;
            printf("scoring inner loop in %d workgroups of %d chunks each:", (int)grid.x, chunks);
            for (int rep = 0; rep < 40; ++rep) {
                (void)hipEventRecord(e0);
                mfma_lds_loop<<<grid, block>>>(out, chunks, 1);
                (void)hipEventRecord(e1);
                if (rep % 10 == 9) {
                    (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    printf(" %.1f", flop / ms / 1e9);
                }
            }
            printf(" TFLOP/s\n");
        }
    }
    // ... and with the kernel's DMA traffic and barriers: long-running workgroups, then one tile per workgroup; tables of 12 MB (L2 / MALL) and 512 MB
    {
        float *out;
        (void)hipMalloc(&out, 4096);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rows : {24576, 1000000}) {
            float *table;
            (void)hipMalloc(&table, (size_t)rows * 128 * 4);
            (void)hipMemset(table, 0x3d, (size_t)rows * 128 * 4);      // (0x3d3d3d3d = 0.046...)
            const char *names[4] = {"DMA + two barriers per chunk (the kernel's structure)", "barriers only, no DMA", "DMA, no barriers (own vmcnt wait)",
                                    "DOUBLE-BUFFERED: next chunk's DMA in flight, one barrier per chunk, 2 workgroups per CU"};
            for (int mode = 0; mode < 4; ++mode) {
                dim3 grid(p.multiProcessorCount * (mode == 3 ? 2 : 4)), block(256);
                const int chunks = mode == 3 ? 1000 : 500;
                const double flop = (double)grid.x * 4 * chunks * 16.0 * 4 * 4096.0;
                printf("table of %d rows x 128, %s:", rows, names[mode]);
                for (int rep = 0; rep < 8; ++rep) {
                    (void)hipEventRecord(e0);
                    if (mode == 0) mfma_dma_loop0<<<grid, block>>>(out, chunks, table, rows, 128);
                    if (mode == 1) mfma_dma_loop1<<<grid, block>>>(out, chunks, table, rows, 128);
                    if (mode == 2) mfma_dma_loop2<<<grid, block>>>(out, chunks, table, rows, 128);
                    if (mode == 3) mfma_dma_loop3<<<grid, block>>>(out, chunks, table, rows, 128);
                    (void)hipEventRecord(e1);
                    if (rep % 2 == 1) {
                        (void)hipEventSynchronize(e1);
                        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                        printf(" %.1f", flop / ms / 1e9);
                    }
                }
                printf(" TFLOP/s\n");
            }
            (void)hipFree(table);
        }
    }
    // workgroup shapes: fewer LDS-DMA instructions per MFMA with larger tiles
    {
        float *out, *table;
        const int rows = 1000000;
        (void)hipMalloc(&out, 4096);
        (void)hipMalloc(&table, (size_t)rows * 128 * 4);
        (void)hipMemset(table, 0x3d, (size_t)rows * 128 * 4);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        struct { const char *name; void (*k)(float *, int, const float *, int, int); int threads, per_cu; size_t lds; } shapes[] = {
            {"2 x 2 wavefronts (128 x 128), 4 workgroups per CU", shape_2x2, 256, 4, (size_t)(128 * 32 + 16) * 2 * 4},
            {"4 x 2 wavefronts (256 x 128), 2 per CU", shape_4x2, 512, 2, (size_t)((256 * 32 + 32) + (128 * 32 + 16)) * 4},
            {"4 x 2, double-buffered, 1-2 per CU", shape_4x2db, 512, 2, (size_t)((256 * 32 + 32) + (128 * 32 + 16)) * 2 * 4},
            {"4 x 4 wavefronts (256 x 256), 1 per CU", shape_4x4, 1024, 1, (size_t)(256 * 32 + 32) * 2 * 4},
            {"4 x 4, double-buffered, 1 per CU", shape_4x4db, 1024, 1, (size_t)(256 * 32 + 32) * 2 * 2 * 4}};
        for (auto &sh : shapes) {
            (void)hipFuncSetAttribute((const void *)sh.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds);
            dim3 grid(p.multiProcessorCount * sh.per_cu), block(sh.threads);
            const int chunks = 500;
            const double flop = (double)grid.x * (sh.threads / 64) * chunks * 16.0 * 4 * 4096.0;
            printf("shape %s (LDS %zu KB per workgroup):", sh.name, sh.lds / 1024);
            for (int rep = 0; rep < 8; ++rep) {
                (void)hipEventRecord(e0);
                sh.k<<<grid, block, sh.lds>>>(out, chunks, table, rows, 128);
                (void)hipEventRecord(e1);
                if (rep % 2 == 1) {
                    (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    printf(" %.1f", flop / ms / 1e9);
                }
            }
            printf(" TFLOP/s  [%s]\n", hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
