// rsx_score.hip -- full-catalog scoring on the matrix cores + seen-item mask +
// per-row Top-K for gfx950 (MI355X).
//
// Restates (no kernel exists in the reference; eager PyTorch / C++ there):
//   models/MF.py:109-112  S = P[users] @ Q.T                     -> score_tile_kernel
//   models/MF.py:130      pred[eval_pos.nonzero()] = -inf        -> mask_seen_kernel
//   evaluation/backend/cython/include/func.h:12-31  partial sort, K best by
//                         descending score, int32 indices          -> topk_rows_kernel
//
// score_tile_kernel: a genuine dense contraction (K = d), so it runs on
// v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain).  128x128 output
// tile per 256-thread workgroup, 2x2 wavefronts of 64x64, 2x2 MFMA tiles of
// 32x32 per wavefront (64 accumulator registers), K staged through LDS in
// chunks of 32 held K-MAJOR ([k][row], leading dimension 129) so that both the
// transposing ds_write_b32 and the fragment ds_read_b32 are bank-conflict free.
#include <mutex>
#include <vector>

#include "rsx_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
#ifndef RSX_SCORE_PREFETCH
#define RSX_SCORE_PREFETCH 1
#endif
// RSX_SCORE_GLDS = 1: the K chunks go from global memory STRAIGHT into LDS (global_load_lds_dwordx4, the LDS-DMA of gfx950)
// instead of through 32 staging registers and 64 transposing ds_write2 per thread and chunk.  The DMA writes lane-linear
// (wave-uniform base + lane * 16 B), so the tiles are ROW-major [row][32] without padding, and the bank conflicts of the
// fragment reads (32 lanes read the same k of 32 consecutive rows: one bank) are broken by an XOR swizzle of the 16-byte slots
// with (row & 7), applied to the per-lane SOURCE address of the DMA and to the read (the same involution on both sides).
// Four wavefronts per SIMD then fit (64 accumulators + 64 other registers; 32.5 KB of LDS per workgroup): round 3, same box,
// 64 x 1024 users x 100K items: 275.1 -> 272.7 us per 1024 users fused, the dense 1024 x 100K product 267 -> 250 us; with three
// wavefronts the DMA form is SLOWER than register staging (286 us: the chunk's load latency is no longer hidden inside the
// workgroup), and register staging compiled for four spills 36 registers (297 us).  GLDS = false remains for item tables of 4 GB
// and more (the DMA's source is a 32-bit offset from the table's base in SGPRs) and as the reference of the A/B.
#ifndef RSX_SCORE_GLDS
#define RSX_SCORE_GLDS 1      // 0: never take the LDS-DMA form (development A/B)
#endif
// RSX_SCORE_LDS_PAD = 1 (round 4): one dword of LDS padding behind every 8 rows of a GLDS tile.  The XOR swizzle alone leaves the 32
// lanes of a fragment read on 8 banks, 4 lanes each (SQ_LDS_BANK_CONFLICT 2.8e8 cycles per pass, a quarter of the LDS's time); a DMA
// instruction writes exactly 8 rows, so its LDS base can carry the pad (the destination is then only dword aligned: the gfx950 DMA
// takes it), and the four 8-row groups a fragment read spans land on four different bank offsets: conflict-free.  Same box, two rounds:
// fused 253.6 / 252.8 -> 246.9 / 248.2 us per 1024 users (0.658 -> 0.673 of the MFMA-fp32 peak), identical Top-50 checksum.
#ifndef RSX_SCORE_LDS_PAD
#define RSX_SCORE_LDS_PAD 1   // 0: the unpadded tiles (development A/B)
#endif
// (s_setprio 2 / 3 around the MFMA block, so that a wavefront in its MFMA phase issues ahead of the others' VALU / LDS work:
//  measured 279.5 -> 283.6 us per 1024 users, same box, round 3 -- dropped.  A FIXED order of precedence among the four wavefronts
//  that share a SIMD -- s_setprio 3 / 2 / 1 / 0 by HW_ID.wave_id, to keep them from reaching the end of a chunk and waiting for
//  the next one all together -- changes nothing: 270.0 / 270.2 vs 270.4 / 270.2 us fused, the dense product 250 -> 256 us)
// (Two LDS buffers per tile pair with the next chunk's DMA in flight under this chunk's MFMAs -- s_waitcnt vmcnt(8) + s_barrier by hand,
//  64 KB of LDS, so TWO workgroups per CU: bit-identical, 256.3 -> 280.5 us per 1024 users.  Four resident wavefronts per SIMD hide
//  the chunk latency better than a prefetch inside two; with 16-wide chunks (four workgroups again) the 64-byte row pitch costs
//  2-way LDS bank conflicts on every fragment read.  Round 3, profiles/r03_exp_scoring_variants.txt.)
// RSX_SCORE_FRAG64 = 1 (round 5): the fragments of TWO MFMA steps come with one ds_read_b64 per tile row instead of two ds_read_b32 -- a
// b64 read takes the same two LDS cycles as a b32 read (MI355X_MICROARCH.md, LDS table) and half the instructions.  The contraction index
// is permuted for that (any order of k is the same sum as long as A and B use the same one): in the pair of steps that covers the k-quad q
// the low half of the wavefront multiplies k = 4q, 4q + 1 and the high half k = 4q + 2, 4q + 3 (it used to be 2 kk / 2 kk + 1 per step).  The
// tiles are swizzled by bits 1-3 of the row and padded by two dwords per 16 rows for it: the 32 lanes of a read then cover the 64 banks once.
#ifndef RSX_SCORE_FRAG64
#define RSX_SCORE_FRAG64 0
#endif
// RSX_SCORE_TILES_PER_WG = T (round 5 experiment): a workgroup computes T consecutive item tiles of its row tile instead of one --
// tools/mfma_peak.hip: the inner loop of this kernel alone sustains 154 TFLOP/s (0.98 of the peak) in long-running workgroups, 136 in
// 50 000 workgroups of one tile each, 143.5 with two: starting and ending workgroups costs the matrix pipe a tenth of its time.
// (PERSISTENT workgroups -- four per CU walking all tiles -- were measured first: 6-7 % SLOWER; they take the hardware's dynamic
//  balancing away and leave the other lane's selection kernels no slot to run in.)
#ifndef RSX_SCORE_TILES_PER_WG
#define RSX_SCORE_TILES_PER_WG 1
#endif
constexpr int kSlots = 2;     // private candidate slots per (64-item strip, row) of the filtered product
constexpr int LDT = BM + 1;   // K-major tile leading dimension (odd -> conflict-free transpose)

// ---------------------------------------------------------------- scoring -------
// FILTER = false: out[row, col] = score (column col is item col*item_stride).
// FILTER = true : nothing is stored densely; every score >= tau[row] is appended to the row's
//                 candidate list (cand_val / cand_idx, capacity cand_cap, counter cand_cnt).
template <int D, bool FILTER, bool GLDS>
__device__ __forceinline__ void score_tile_body(const float *__restrict__ P,
                                                         const int32_t *__restrict__ user_ids,
                                                         int64_t num_rows,
                                                         const float *__restrict__ Q,
                                                         int64_t num_items, int64_t item_stride,
                                                         float *__restrict__ out,
                                                         const float *__restrict__ tau,
                                                         float *__restrict__ cand_val,
                                                         int32_t *__restrict__ cand_idx,
                                                         int32_t *__restrict__ cand_cnt, int cand_cap,
                                                         uint2 *__restrict__ slots, int64_t item_base,
                                                         int64_t n_item_tiles, int64_t n_tiles)
{
    // (GLDS + RSX_SCORE_LDS_PAD: one dword of padding behind every 8 rows -- the 8 rows one DMA instruction of a wavefront writes.  With
    //  the XOR swizzle alone the 32 lanes of a fragment read fall on 8 banks, 4 lanes each; the pad moves the four 8-row groups of
    //  those lanes to four different bank offsets: conflict-free.)
    constexpr int PADA = (GLDS && RSX_SCORE_FRAG64) ? 2 * (BM / 16) : (GLDS && RSX_SCORE_LDS_PAD) ? BM / 8 : 0;
    __shared__ __attribute__((aligned(16))) float As[GLDS ? BM * BK + PADA : BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[GLDS ? BN * BK + PADA : BK * LDT];
    __shared__ __attribute__((aligned(16))) float tau_s2[2][BM];    // (two: a persistent workgroup's next tile writes its thresholds while slower wavefronts still read this tile's)

    // item tile varies fastest: the 8 XCDs each stream different item tiles of
    // the SAME user tile, and a user tile's A panel (128 x d) stays L2 resident.
    // this workgroup's RSX_SCORE_TILES_PER_WG consecutive item tiles of one row tile (one pass, known at compile time, when that is 1)
    const uint32_t n_it = (uint32_t)n_item_tiles;
    int tile_par = 0;
    uint32_t tile_x = blockIdx.x * RSX_SCORE_TILES_PER_WG;
    const uint32_t tile_x_end = (tile_x + RSX_SCORE_TILES_PER_WG < n_it) ? tile_x + RSX_SCORE_TILES_PER_WG : n_it;
    const uint32_t tile_y = blockIdx.y;
    do {
    float *const tau_s = tau_s2[tile_par];
    // (a workgroup of several tiles recomputes its per-thread constants for every tile -- a dozen vector instructions -- instead of carrying
    //  them, hoisted by the compiler, across the tile loop: carried, they cost the 128-register budget of four wavefronts per SIMD 47-89 spills)
    int tid = threadIdx.x;
    if (RSX_SCORE_TILES_PER_WG > 1) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int64_t item0 = (int64_t)tile_x * BN;
    const int64_t row0 = (int64_t)tile_y * BM;

    // staging assignment: thread -> (row = tid>>3 + 32 n, k-quad = tid&7)
    const int kq = tid & 7;
    const int srow = tid >> 3;
    const float *a_src[4];
    const float *b_src[4];
    uint32_t b_off[4];                 // GLDS: byte offsets of the item rows this thread stages (item table below 4 GB: the launcher checks)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        // rows / items past the edge are clamped to the last valid one: their products are
        // computed and never stored, which keeps every staging load an unconditional dwordx4
        int64_t r = row0 + srow + 32 * n;
        r = (r < num_rows) ? r : num_rows - 1;
        // GLDS: LDS slot kq of this row receives the row's k-quad kq ^ (row & 7)  (FRAG64: kq ^ ((row >> 1) & 7))
        const int gq = GLDS ? (kq ^ (RSX_SCORE_FRAG64 ? ((srow >> 1) & 7) : (srow & 7))) : kq;
        int64_t it = item0 + srow + 32 * n;
        it = (it < num_items) ? it : num_items - 1;
        a_src[n] = P + (size_t)user_ids[r] * D + 4 * gq;
        b_src[n] = Q + (size_t)(it * item_stride) * D + 4 * gq;
        b_off[n] = (uint32_t)(it * item_stride) * (uint32_t)(D * 4) + (uint32_t)(16 * gq);
    }
    if constexpr (FILTER) {
        // (dev build only, tests/test_mutation.py: mask 16 lifts a positive threshold by a quarter -- true Top-K items are filtered out)
        if (tid < BM) tau_s[tid] = (row0 + tid < num_rows) ? tau[row0 + tid] * ((RSX_ABL(16) && tau[row0 + tid] > 0.f) ? 1.25f : 1.0f) : INFINITY;
    }
    const int k_end = RSX_ABL(8) ? D - BK : D;      // (dev build only: mask 8 drops the last K chunk of the product)

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    if constexpr (GLDS && RSX_SCORE_FRAG64) {
    // lane's fragment rows: wr * 64 + l31 (+ 32) of A, wc * 64 + l31 (+ 32) of B.  Element (row, k) sits at dword
    //   row * 32 + (((k >> 2) ^ ((row >> 1) & 7)) << 2) + (k & 3) + 2 * (row >> 4)
    // and this lane reads the two floats k = 4 q + 2 hi, 4 q + 2 hi + 1 of the quad q: one ds_read_b64 per row and pair of steps
    const int xq = (l31 >> 1) & 7;                      // the same for row and row + 32 (and for wr * 64 + ..., wc * 64 + ...)
    const float *ap = As + (wr * 64 + l31) * BK + ((wr * 64 + l31) >> 4) * 2 + 2 * hi;
    const float *bp = Bs + (wc * 64 + l31) * BK + ((wc * 64 + l31) >> 4) * 2 + 2 * hi;
    constexpr int kRow32 = 32 * BK + 4;                 // rows r and r + 32: two 16-row pads apart
    auto frag = [&](const float *base, int q) __attribute__((always_inline)) { return *reinterpret_cast<const float2 *>(base + ((q ^ xq) << 2)); };
    for (int k0 = 0; k0 < k_end; k0 += BK) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float *gb = reinterpret_cast<const float *>(reinterpret_cast<const char *>(Q) + (b_off[n] + (uint32_t)k0 * 4u));
            __builtin_amdgcn_global_load_lds(a_src[n] + k0, As + (32 * n + 8 * wid) * BK + (2 * n + (wid >> 1)) * 2, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gb, Bs + (32 * n + 8 * wid) * BK + (2 * n + (wid >> 1)) * 2, 16, 0, 0);
        }
        __syncthreads();          // (carries the vmcnt(0) of the DMA: the chunk has landed for every wavefront)
        float2 a0 = frag(ap, 0), a1 = frag(ap + kRow32, 0), b0 = frag(bp, 0), b1 = frag(bp + kRow32, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
            float2 na0 = make_float2(0.f, 0.f), na1 = na0, nb0 = na0, nb1 = na0;
            if (q + 1 < BK / 4) { na0 = frag(ap, q + 1); na1 = frag(ap + kRow32, q + 1); nb0 = frag(bp, q + 1); nb1 = frag(bp + kRow32, q + 1); }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[1][1], 0, 0, 0);
            // the four reads of the NEXT pair of steps, then this pair's eight MFMAs (the reads are a whole pair ahead of their use)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        __syncthreads();          // every wavefront is done reading before the next chunk overwrites the tiles
    }
    } else if constexpr (GLDS) {
    // lane's fragment rows: wr * 64 + l31 (+ 32) of A, wc * 64 + l31 (+ 32) of B; element (row, k) sits at row * 32 + (k ^ x),
    // x = (row & 7) << 2 = (l31 & 7) << 2 for all four rows; this lane reads k = kk + hi
    const int xs = (l31 & 7) << 2;
    constexpr int P8 = RSX_SCORE_LDS_PAD ? 1 : 0;      // dwords of padding per 8-row group
    const float *ap = As + (wr * 64 + l31) * BK + ((wr * 64 + l31) >> 3) * P8 + hi;
    const float *bp = Bs + (wc * 64 + l31) * BK + ((wc * 64 + l31) >> 3) * P8 + hi;
    for (int k0 = 0; k0 < k_end; k0 += BK) {
        // 8 rows x 128 B per wave instruction, rows 32 n + 8 wid .. + 8 of each tile
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            // (item rows: 32-bit byte offsets from the table's base, which stays in SGPRs; user rows: pointers, the user table may be larger)
            const float *gb = reinterpret_cast<const float *>(reinterpret_cast<const char *>(Q) + (b_off[n] + (uint32_t)k0 * 4u));
            __builtin_amdgcn_global_load_lds(a_src[n] + k0, As + (32 * n + 8 * wid) * BK + (4 * n + wid) * P8, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gb, Bs + (32 * n + 8 * wid) * BK + (4 * n + wid) * P8, 16, 0, 0);
        }
        __syncthreads();          // (carries the vmcnt(0) of the DMA: the chunk has landed for every wavefront)
        float a0 = ap[xs], a1 = ap[32 * BK + 4 * P8 + xs], b0 = bp[xs], b1 = bp[32 * BK + 4 * P8 + xs];
        // (the fragments of a step are FOUR ds_read_b32 with the padded tiles -- rows r and r + 32 are 4 096 + 16 bytes apart, which no
        //  ds_read2 form can address -- and two ds_read2st64_b32 without: the group sizes below must say so, or the reads of step k + 1
        //  drift behind the MFMAs of step k and are waited for at once)
        constexpr int kDsPerStep = RSX_SCORE_LDS_PAD ? 4 : 2;
        __builtin_amdgcn_sched_group_barrier(0x100, kDsPerStep, 0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < BK) {
                const int o = (kk + 2) ^ xs;
                na0 = ap[o]; na1 = ap[32 * BK + 4 * P8 + o];
                nb0 = bp[o]; nb1 = bp[32 * BK + 4 * P8 + o];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, kDsPerStep, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        __syncthreads();          // every wavefront is done reading before the next chunk overwrites the tiles
    }
    } else {
    float4 ra[4], rb[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        ra[n] = *reinterpret_cast<const float4 *>(a_src[n]);
        rb[n] = *reinterpret_cast<const float4 *>(b_src[n]);
    }

    for (int k0 = 0; k0 < k_end; k0 += BK) {
        // registers -> LDS, transposed to K-major
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int r = srow + 32 * n;
            As[(4 * kq + 0) * LDT + r] = ra[n].x; As[(4 * kq + 1) * LDT + r] = ra[n].y;
            As[(4 * kq + 2) * LDT + r] = ra[n].z; As[(4 * kq + 3) * LDT + r] = ra[n].w;
            Bs[(4 * kq + 0) * LDT + r] = rb[n].x; Bs[(4 * kq + 1) * LDT + r] = rb[n].y;
            Bs[(4 * kq + 2) * LDT + r] = rb[n].z; Bs[(4 * kq + 3) * LDT + r] = rb[n].w;
        }
        __syncthreads();
        // issue the next chunk's global loads; they fly under the MFMAs below
        if (k0 + BK < k_end) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                ra[n] = *reinterpret_cast<const float4 *>(a_src[n] + k0 + BK);
                rb[n] = *reinterpret_cast<const float4 *>(b_src[n] + k0 + BK);
            }
        }
        const float *ap = As + hi * LDT + wr * 64 + l31;
        const float *bp = Bs + hi * LDT + wc * 64 + l31;
#if RSX_SCORE_PREFETCH
        // The fragments of step kk + 2 are read from LDS BEFORE the four MFMAs of step kk are issued (two register sets).
        // Left to itself the compiler reuses one set: it issues the next ds_read2 pair only after the fourth MFMA and waits
        // for it at once -- an LDS round trip exposed behind every 256 cycles of matrix work (round 3: the disassembly showed
        // `ds_read2 x2; s_waitcnt lgkmcnt(0); v_mfma x4` sixteen times per chunk).  The sched_group_barriers pin the order.
        float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // step 0's fragments
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < BK) {
                na0 = ap[(kk + 2) * LDT]; na1 = ap[(kk + 2) * LDT + 32];
                nb0 = bp[(kk + 2) * LDT]; nb1 = bp[(kk + 2) * LDT + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // the two ds_read2 of the NEXT step ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // ... then this step's four MFMAs
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
#else
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = ap[kk * LDT], a1 = ap[kk * LDT + 32];
            const float b0 = bp[kk * LDT], b1 = bp[kk * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#endif
        __syncthreads();
    }

    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if constexpr (FILTER) {
        // No barrier, no LDS, no global atomic on the common path.  For one accumulator register
        // (m, r) the 32 lanes of a half-wave hold 2 x 32 consecutive columns of ONE row, so the
        // survivors' ranks inside this wavefront's 64-column strip come from two ballots and a
        // popcount.  Each (64-column strip, row) cell owns kSlots private slots; its count byte is
        // written by the half-wave's first lane.  (History, all measured on a 1024 x 100K tile:
        // a returning global atomic per hit 909 us; LDS compaction + per-row global counters
        // 560 us -- 782 item tiles x 8 XCDs on 1024 counters cost ~0.5 us per atomic; LDS
        // compaction + private cells 345 us; this form: see DESIGN.md.)
        // the slot array is ROW-major [row][strip][kSlots] (the merge kernel reads one contiguous
        // run per row) and 0xFF-filled before the launch: an unused slot keeps item index -1, so
        // no per-cell count has to be written.  32-bit cell arithmetic, one base per wavefront.
        const int n_strips = (int)n_item_tiles * 2;
        const int strip = (int)tile_x * 2 + wc;                                  // 64-column strip id
        // (round 3: the hit path in 32-bit arithmetic -- with ~29 survivors per wavefront and tile more than half of the 32 sites
        //  take it, which makes it the largest block of non-MFMA vector instructions of the kernel: ranks from mbcnt instead of
        //  two 64-bit and + popcount pairs, the slot addressed by a 32-bit byte offset from the array's base in SGPRs)
        const uint32_t lomask = hi ? 0u : 0xFFFFFFFFu;                            // a high-half lane counts survivors of its own half only
        const int32_t col0 = (int32_t)item0 + wc * 64 + l31, col1 = col0 + 32;    // (catalogs fit int32)
        const int wave_row = (int)row0 + wr * 64 + 4 * hi;                       // + per-site constant
        const uint32_t wave_off = (uint32_t)((wave_row * n_strips + strip) * kSlots) * 8u;    // bytes into `slots` (< 4 GB: rsx_score_topk_workspace)
        // (recording the survivors in registers and storing them after the walk was tried: the
        //  fully unrolled walk then needs 126 VGPRs and the whole kernel slows down by 15 %)
        // lanes whose column exists (the last item tile is ragged), as wave masks: ANDed into the survivor masks below
        const unsigned long long cm0 = RSX_ABL(4) ? 0ull : __builtin_amdgcn_ballot_w64(col0 < (int32_t)num_items);
        const unsigned long long cm1 = RSX_ABL(4) ? 0ull : __builtin_amdgcn_ballot_w64(col1 < (int32_t)num_items);
        // (round 3) the thresholds of four consecutive sites -- rows (r & 3) = 0..3 of one group of eight -- come with ONE
        // ds_read_b128 instead of four ds_read_b32 each waited for at once, and the survivor masks are taken straight from the
        // compares (__builtin_amdgcn_ballot_w64; __ballot went through v_cndmask + v_cmp_ne per mask): 8 -> 4 vector
        // instructions per site on the path where nothing survives, which is nearly every site
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int site = m * 32 + (r & 3) + 8 * (r >> 2);                // compile-time constant
                float4 t4;
                if ((r & 3) == 0) t4 = *reinterpret_cast<const float4 *>(&tau_s[wr * 64 + 4 * hi + site]);   // sites r .. r + 3
                const float t = (r & 3) == 0 ? t4.x : (r & 3) == 1 ? t4.y : (r & 3) == 2 ? t4.z : t4.w;      // +inf for rows past the edge
                const float v0 = acc[m][0][r], v1 = acc[m][1][r];
                const unsigned long long b0 = __builtin_amdgcn_ballot_w64(v0 >= t) & cm0, b1 = __builtin_amdgcn_ballot_w64(v1 >= t) & cm1;
                if ((b0 | b1) == 0ull) continue;                                  // wave-uniform
                const uint32_t b0lo = (uint32_t)b0, b0hi = (uint32_t)(b0 >> 32), b1lo = (uint32_t)b1, b1hi = (uint32_t)(b1 >> 32);
                const uint32_t cnt0 = hi ? (uint32_t)__builtin_popcount(b0hi) : (uint32_t)__builtin_popcount(b0lo);
                const uint32_t site_off = (uint32_t)(site * n_strips * kSlots) * 8u;
                auto emit = [&](uint32_t rank, float v, int32_t col) {
                    if (rank < (uint32_t)kSlots) {
                        uint2 *cell = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(slots) + (wave_off + site_off + rank * 8u));
                        *cell = make_uint2(__float_as_uint(v), (unsigned)col + (unsigned)item_base);
                    } else {                            // rare: more than kSlots survivors in one cell
                        const int64_t row = wave_row + site;
                        const int slot = atomicAdd(cand_cnt + row, 1);
                        if (slot < cand_cap) {
                            cand_val[(size_t)row * cand_cap + slot] = v;
                            cand_idx[(size_t)row * cand_cap + slot] = (int32_t)((unsigned)col + (unsigned)item_base);
                        }
                    }
                };
                if (!RSX_ABL(2)) {
                    if ((v0 >= t) && col0 < (int32_t)num_items)
                        emit(__builtin_amdgcn_mbcnt_hi(b0hi, __builtin_amdgcn_mbcnt_lo(b0lo & lomask, 0u)), v0, col0);
                    if ((v1 >= t) && col1 < (int32_t)num_items)
                        emit(cnt0 + __builtin_amdgcn_mbcnt_hi(b1hi, __builtin_amdgcn_mbcnt_lo(b1lo & lomask, 0u)), v1, col1);
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int64_t row = row0 + lrow;
                if (row >= num_rows) continue;
                float *orow = out + (size_t)row * num_items;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int64_t col = item0 + wc * 64 + n * 32 + l31;
                    if (col < num_items) orow[col] = acc[m][n][r];
                }
            }
        }
    }
    ++tile_x; tile_par ^= 1;
    } while (RSX_SCORE_TILES_PER_WG > 1 && tile_x < tile_x_end);
}

// the two entry points: register staging (three wavefronts per SIMD) and LDS-DMA (four).  (One template with a launch bound that
// depends on GLDS left the host stubs of the GLDS = true instantiations undefined at link time: two kernels, one body.)
#define RSX_SCORE_ARGS const float *__restrict__ P, const int32_t *__restrict__ user_ids, int64_t num_rows, const float *__restrict__ Q, \
                       int64_t num_items, int64_t item_stride, float *__restrict__ out, const float *__restrict__ tau,               \
                       float *__restrict__ cand_val, int32_t *__restrict__ cand_idx, int32_t *__restrict__ cand_cnt, int cand_cap,  \
                       uint2 *__restrict__ slots, int64_t item_base, int64_t n_item_tiles, int64_t n_tiles
#define RSX_SCORE_PASS P, user_ids, num_rows, Q, num_items, item_stride, out, tau, cand_val, cand_idx, cand_cnt, cand_cap, slots, item_base, n_item_tiles, n_tiles
template <int D, bool FILTER>
__global__ __launch_bounds__(256) void score_tile_kernel(RSX_SCORE_ARGS) { score_tile_body<D, FILTER, false>(RSX_SCORE_PASS); }
template <int D, bool FILTER>
__global__ __launch_bounds__(256, 4) void score_tile_glds_kernel(RSX_SCORE_ARGS) { score_tile_body<D, FILTER, true>(RSX_SCORE_PASS); }
#undef RSX_SCORE_ARGS
#undef RSX_SCORE_PASS

// scores[r, indices[p]] = -inf for p in the CSR row of user_ids[r]
__global__ __launch_bounds__(256) void mask_seen_kernel(float *__restrict__ scores,
                                                        const int32_t *__restrict__ user_ids,
                                                        int64_t num_rows, int64_t num_items,
                                                        const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices)
{
    const int64_t r = blockIdx.x;
    if (r >= num_rows) return;
    const int32_t u = user_ids[r];
    const int64_t lo = indptr[u], hi = indptr[u + 1];
    float *row = scores + (size_t)r * num_items;
    for (int64_t p = lo + threadIdx.x; p < hi; p += blockDim.x) row[indices[p]] = -INFINITY;
}

// ---------------------------------------------------------------- top-k ----------
// order-preserving float -> uint32 (larger float <=> larger key; -inf smallest)
__device__ __forceinline__ uint32_t f2key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

constexpr int TK_THREADS = 256;
constexpr int TK_BINS = 4096;
constexpr int TK_CAP = 2048;   // candidate list capacity (>= 2 * max K)

// One workgroup per row.  Radix select on the key (12 + 12 + 8 bits) until the
// candidates (everything not provably below the K-th key) fit the LDS list,
// then a bitonic sort of that list by (key desc, index asc).
__global__ __launch_bounds__(TK_THREADS) void topk_rows_kernel(const float *__restrict__ scores,
                                                               int64_t num_items, int K,
                                                               int32_t *__restrict__ out_idx,
                                                               float *__restrict__ out_val)
{
    __shared__ uint32_t hist[TK_BINS];
    __shared__ unsigned long long cand[TK_CAP];
    __shared__ uint32_t s_wave[TK_THREADS / 64];
    __shared__ uint32_t s_bin, s_need, s_count, s_ncand, s_run;

    const int tid = threadIdx.x;
    const float *row = scores + (size_t)blockIdx.x * num_items;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(row) % 16 == 0) ? num_items / 4 : 0;  // vector part

    uint32_t prefix = 0;       // high bits of the K-th largest key fixed so far
    uint32_t need = (uint32_t)K;  // rank still to resolve inside the prefix class
    int shift = 32;
    bool gathered = false;
    const int level_bits[3] = {12, 12, 8};

    for (int level = 0; level < 3 && !gathered; ++level) {
        const int bits = level_bits[level];
        const int nshift = shift - bits;
        const uint32_t bmask = (1u << bits) - 1u;
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
        auto tally = [&](float f) {
            const uint32_t key = f2key(f);
            if (shift == 32 || (key >> shift) == (prefix >> shift)) atomicAdd(&hist[(key >> nshift) & bmask], 1u);
        };
        for (int64_t q = tid; q < n4; q += TK_THREADS) {
            const float4 v = reinterpret_cast<const float4 *>(row)[q];
            tally(v.x); tally(v.y); tally(v.z); tally(v.w);
        }
        for (int64_t c = n4 * 4 + tid; c < num_items; c += TK_THREADS) tally(row[c]);
        __syncthreads();
        // find the bin holding the need-th largest: suffix counts from the top bin down.
        // thread t owns bins [t*16, t*16+16) (only the first (1<<bits)/16 threads have bins)
        const int per = TK_BINS / TK_THREADS;
        uint32_t mine = 0;
        for (int b = 0; b < per; ++b) mine += hist[tid * per + b];
        // inclusive suffix sum over threads (higher tid = higher bins)
        uint32_t suf = mine;
        {
            const int lane = tid & 63;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_down(suf, o, 64);
                if (lane + o < 64) suf += t;
            }
            if (lane == 0) s_wave[tid >> 6] = suf;
            __syncthreads();
            for (int w = (tid >> 6) + 1; w < TK_THREADS / 64; ++w) suf += s_wave[w];
        }
        const uint32_t above = suf - mine;   // elements in bins of higher threads
        if (above < need && suf >= need) {   // the wanted bin is one of mine
            uint32_t run = above;
            for (int b = per - 1; b >= 0; --b) {
                const uint32_t h = hist[tid * per + b];
                if (run + h >= need) {
                    s_bin = (uint32_t)(tid * per + b);
                    s_need = need - run;     // rank inside that bin
                    s_count = h;
                    break;
                }
                run += h;
            }
        }
        __syncthreads();
        const uint32_t bin = s_bin, cnt = s_count;
        const uint32_t sure = (uint32_t)K - s_need;   // strictly above the bin: certainly selected
        need = s_need;
        prefix |= bin << nshift;
        shift = nshift;
        if (sure + cnt <= (uint32_t)TK_CAP || level == 2) {
            // gather every element whose high bits are >= prefix's (>, or == when the
            // tie class fits; a too-large exact-tie class is resolved in index order below)
            const bool take_ties = (sure + cnt <= (uint32_t)TK_CAP);
            if (tid == 0) { s_ncand = 0; s_run = 0; }
            __syncthreads();
            auto consider = [&](float f, int64_t c) {
                const uint32_t key = f2key(f);
                const uint32_t hk = key >> shift, hp = prefix >> shift;
                if (hk > hp || (take_ties && hk == hp)) {
                    const uint32_t slot = atomicAdd(&s_ncand, 1u);
                    cand[slot] = ((unsigned long long)key << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)c);
                }
            };
            for (int64_t q = tid; q < n4; q += TK_THREADS) {
                const float4 v = reinterpret_cast<const float4 *>(row)[q];
                consider(v.x, 4 * q); consider(v.y, 4 * q + 1); consider(v.z, 4 * q + 2); consider(v.w, 4 * q + 3);
            }
            for (int64_t c = n4 * 4 + tid; c < num_items; c += TK_THREADS) consider(row[c], c);
            __syncthreads();
            if (!take_ties) {
                // huge class of EXACTLY tied keys (shift == 0 here): take the `need`
                // lowest indices, walking the row in index order.
                for (int64_t base = 0; base < num_items && s_run < need; base += TK_THREADS) {
                    const int64_t c = base + tid;
                    const bool tie = (c < num_items) && (f2key(row[c]) == prefix);
                    const unsigned long long bal = __ballot(tie);
                    const int lane = tid & 63;
                    if (lane == 0) s_wave[tid >> 6] = (uint32_t)__popcll(bal);
                    __syncthreads();
                    uint32_t before = s_run;
                    for (int w = 0; w < (tid >> 6); ++w) before += s_wave[w];
                    before += (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (tie && before < need) {
                        const uint32_t slot = atomicAdd(&s_ncand, 1u);
                        cand[slot] = ((unsigned long long)prefix << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)c);
                    }
                    __syncthreads();
                    if (tid == 0) {
                        uint32_t tot = 0;
                        for (int w = 0; w < TK_THREADS / 64; ++w) tot += s_wave[w];
                        s_run += tot;
                    }
                    __syncthreads();
                }
            }
            gathered = true;
        }
        __syncthreads();
    }

    // bitonic sort of the candidate list, descending on the 64-bit (key, ~index) word
    const uint32_t ncand = s_ncand;
    uint32_t n2 = 1;
    while (n2 < ncand) n2 <<= 1;
    for (uint32_t t = ncand + tid; t < n2; t += TK_THREADS) cand[t] = 0ull;   // sorts last
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < n2 / 2; t += TK_THREADS) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi2 = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi2];
                if ((a < b) == desc) { cand[lo] = b; cand[hi2] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < K; t += TK_THREADS) {
        // (dev build only, tests/test_mutation.py: mask 64 hands out the (K+1)-th candidate in the K-th place)
        const unsigned long long e = cand[(RSX_ABL(64) && t == K - 1 && (uint32_t)K < ncand) ? t + 1 : t];
        out_idx[(size_t)blockIdx.x * K + t] = (t < (int)ncand) ? (int32_t)(0xFFFFFFFFu - (uint32_t)e) : -1;
        if (out_val) out_val[(size_t)blockIdx.x * K + t] = (t < (int)ncand) ? key2f((uint32_t)(e >> 32)) : -INFINITY;
    }
}

// ---- fused scoring + Top-K (no dense score matrix) ------------------------------------------
// 1. the item table is copied in the order p -> (a p) mod I; the FIRST kSampleCols rows of the copy -- an equidistributed
//    sample of the catalog whatever the ids mean -- are scored densely, and sample_tau_kernel takes the exact K-th value
//    tau[row] of the unseen ones (a lower bound of the row's final K-th score) and emits those K as candidates;
// 2. the REST of the copy is scored with the FILTER epilogue: only scores >= tau[row] leave the CU (every item is scored
//    exactly once);
// 3. per row: map the copy's row numbers back to item ids, drop seen items, bitonic sort, emit K.
// A row whose candidates overflow the list (massive exact ties) is redone through the dense path.

// tau of the fused path in ONE pass over the sample scores: the K-th largest value of a row of the [rows x kSampleCols]
// sample product (the first kSampleCols rows of the PERMUTED item table) after the seen items are masked out.  Instead of a mask pass + a sorted top-K + a gather of its K-th column (which read the
// 268 MB of sample scores twice and sorted candidates nobody needs in order): each of the 256 threads keeps
// its 32 keys of the row in registers, the seen sample columns are a 1 KB bitmap in LDS, and three radix levels
// (12 + 12 + 8 bits, LDS histogram) fix the K-th key exactly.  Same value as the K-th entry of the sorted top-K.
#ifndef RSX_SAMPLE_COLS
#define RSX_SAMPLE_COLS 8192
#endif
constexpr int ST_COLS = RSX_SAMPLE_COLS;
constexpr int ST_NPT = ST_COLS / TK_THREADS;     // 32 keys per thread
__global__ __launch_bounds__(TK_THREADS) void sample_tau_kernel(const float *__restrict__ sample,
                                                                const int32_t *__restrict__ user_ids, int64_t num_rows,
                                                                int64_t perm_inv, int64_t perm_n,
                                                                const int64_t *__restrict__ indptr,
                                                                const int32_t *__restrict__ indices, int K,
                                                                float *__restrict__ tau, int32_t *__restrict__ cand_cnt,
                                                                float *__restrict__ cand_val, int32_t *__restrict__ cand_idx,
                                                                int cand_cap)
{
    __shared__ uint32_t hist[TK_BINS];
    __shared__ uint32_t s_emit;
    __shared__ uint32_t seen[ST_COLS / 32];
    __shared__ uint32_t s_wave[TK_THREADS / 64];
    __shared__ uint32_t s_bin, s_need;
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const float *row = sample + (size_t)r * ST_COLS;
    uint32_t key[ST_NPT];
#pragma unroll
    for (int e = 0; e < ST_NPT; ++e) key[e] = f2key(row[e * TK_THREADS + tid]);
    if (indptr != nullptr) {
        for (int q = tid; q < ST_COLS / 32; q += TK_THREADS) seen[q] = 0u;
        __syncthreads();
        const int32_t u = user_ids[r];
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        for (int64_t p = lo + tid; p < hi; p += TK_THREADS) {
            const int64_t col = ((int64_t)indices[p] * perm_inv) % perm_n;      // where the permuted table holds this item
            if (col < ST_COLS) atomicOr(&seen[col >> 5], 1u << (col & 31));
        }
        __syncthreads();
        const uint32_t kinf = f2key(-INFINITY);
#pragma unroll
        for (int e = 0; e < ST_NPT; ++e) {
            const int col = e * TK_THREADS + tid;
            if ((seen[col >> 5] >> (col & 31)) & 1u) key[e] = kinf;
        }
    }
    uint32_t prefix = 0, need = (uint32_t)K;
    int shift = 32;
    const int level_bits[3] = {12, 12, 8};
    for (int level = 0; level < 3; ++level) {
        const int bits = level_bits[level];
        const int nshift = shift - bits;
        const uint32_t bmask = (1u << bits) - 1u;
        for (int b = tid; b < TK_BINS; b += TK_THREADS) hist[b] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < ST_NPT; ++e)
            if (shift == 32 || (key[e] >> shift) == (prefix >> shift)) atomicAdd(&hist[(key[e] >> nshift) & bmask], 1u);
        __syncthreads();
        const int per = TK_BINS / TK_THREADS;
        uint32_t mine = 0;
        for (int b = 0; b < per; ++b) mine += hist[tid * per + b];
        uint32_t suf = mine;
        {
            const int lane = tid & 63;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_down(suf, o, 64);
                if (lane + o < 64) suf += t;
            }
            if (lane == 0) s_wave[tid >> 6] = suf;
            __syncthreads();
            for (int w = (tid >> 6) + 1; w < TK_THREADS / 64; ++w) suf += s_wave[w];
        }
        const uint32_t above = suf - mine;
        if (above < need && suf >= need) {
            uint32_t run = above;
            for (int b = per - 1; b >= 0; --b) {
                const uint32_t h = hist[tid * per + b];
                if (run + h >= need) { s_bin = (uint32_t)(tid * per + b); s_need = need - run; break; }
                run += h;
            }
        }
        __syncthreads();
        need = s_need;
        prefix |= s_bin << nshift;
        shift = nshift;
        __syncthreads();
    }
    // The sample IS part of the catalog (its first ST_COLS items when stride == 1): its own candidates -- every
    // unseen sample item at or above the K-th key -- go to the row's spill list, and the filtered product then covers
    // only the REST of the catalog.  More than the list holds (massive exact ties): the count says so and the merge
    // sends the row to the dense re-do.
    if (tid == 0) s_emit = 0u;
    __syncthreads();
    if (cand_val != nullptr) {
#pragma unroll
        for (int e = 0; e < ST_NPT; ++e) {
            if (key[e] >= prefix) {
                const uint32_t at = atomicAdd(&s_emit, 1u);
                if (at < (uint32_t)cand_cap) {
                    cand_val[(size_t)r * cand_cap + at] = key2f(key[e]);
                    cand_idx[(size_t)r * cand_cap + at] = (int32_t)(e * TK_THREADS + tid);      // permuted id
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) { tau[r] = key2f(prefix); cand_cnt[r] = (int32_t)s_emit; }
}

// Qp[p] = Q[(a p) mod I]: the item table in the order p -> a p mod I (a coprime to I, a ~ I / kSampleCols), so that the
// FIRST kSampleCols rows of Qp are an equidistributed sample of the catalog whatever the item ids mean (a contiguous
// prefix of Q itself would be a biased sample wherever ids correlate with popularity: a loose tau, a flood of survivors)
__global__ __launch_bounds__(256) void permute_items_kernel(const float4 *__restrict__ Q, float4 *__restrict__ Qp,
                                                            int64_t num_items, int d4, int64_t perm_a)
{
    const int64_t n = num_items * d4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int64_t p = t / d4, c = t - p * d4;
        Qp[t] = Q[((p * perm_a) % num_items) * d4 + c];
    }
}

constexpr int MG_THREADS = 256;
constexpr int MG_CAP = 4096;    // candidate list capacity per row.  (Since round 4 the survivors enter the list RAW -- seen items included, they are
                                // dropped after the histogram cut -- so a user whose seen items crowd the top of the catalog overflows, and takes
                                // the dense re-do, a little earlier than when seen items were filtered on arrival: a cost, never a wrong row.)
constexpr int MG_SEEN = 512;    // seen items of a user held in LDS for the mask test (longer rows: searched in the CSR)
constexpr int MG_BATCH = 8;     // cells a thread requests together
constexpr int MG_BIN_BITS = 10;
constexpr int MG_BINS = 1 << MG_BIN_BITS;

__global__ __launch_bounds__(MG_THREADS) void merge_candidates_kernel(
    const uint2 *__restrict__ slots, int64_t n_strips, int64_t tile_rows,
    const float *__restrict__ cand_val, const int32_t *__restrict__ cand_idx,
    const int32_t *__restrict__ cand_cnt, int cand_cap, const int32_t *__restrict__ user_ids,
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int K,
    int32_t *__restrict__ out_idx, float *__restrict__ out_val, int64_t row_base,
    int32_t *__restrict__ overflow_rows, int32_t *__restrict__ overflow_count, int64_t perm_a, int64_t perm_n)
{
    __shared__ unsigned long long cand[MG_CAP];
    __shared__ int32_t seen[MG_SEEN];           // the user's seen items (sorted CSR row), when they fit
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    if (tid == 0) s_n = 0;
    const int spill = cand_cnt[row];
    if (spill > cand_cap) {                     // block-uniform: the spill list itself overflowed
        if (tid == 0) overflow_rows[atomicAdd(overflow_count, 1)] = (int32_t)(row_base + row);
        return;
    }
    int64_t lo = 0, hi = 0;
    if (indptr != nullptr) { const int32_t u = user_ids[row]; lo = indptr[u]; hi = indptr[u + 1]; }
    const int deg = (int)(hi - lo);
    const bool seen_lds = hi - lo <= (int64_t)MG_SEEN;
    if (seen_lds)
        for (int t = tid; t < deg; t += MG_THREADS) seen[t] = indices[lo + t];
    __syncthreads();
    // ONE pass over the row's cells (round 3; it used to count first and fetch again); the cells of a batch are requested together
    // (MG_BATCH loads in flight per thread instead of one after the other).
    // Round 4: the survivors go to the list RAW -- (key, id in the permuted table) -- and the three expensive things a survivor used
    // to pay on arrival (the 64-bit modulo that maps its id back, the search among the user's seen items, its place in the sort) are
    // paid only by those that can still be among the K best: a histogram of the keys finds the bin of the (K + deg)-th largest --
    // at most deg of the entries above it are seen items, so at least K unseen ones are -- and everything below that bin is dropped
    // unexamined (~560 of a row's ~660 survivors at the bench shape).  A list that would not fit sends the row to the dense re-do.
    const uint2 *rslots = slots + (size_t)row * n_strips * kSlots;     // this row's cells, contiguous
    const int64_t n_slots = n_strips * kSlots;  // an unused slot still holds the 0xFF fill: item < 0
    auto push_raw = [&](float v, int32_t pit) {
        const uint32_t at = atomicAdd(&s_n, 1u);
        if (at < (uint32_t)MG_CAP) cand[at] = ((unsigned long long)f2key(v) << 32) | (uint32_t)pit;
    };
    for (int64_t q0 = 0; q0 < n_slots; q0 += MG_THREADS * MG_BATCH) {
        uint2 e[MG_BATCH];
#pragma unroll
        for (int b_ = 0; b_ < MG_BATCH; ++b_) {
            const int64_t q = q0 + b_ * MG_THREADS + tid;
            e[b_] = q < n_slots ? rslots[q] : make_uint2(0u, 0xFFFFFFFFu);
        }
#pragma unroll
        for (int b_ = 0; b_ < MG_BATCH; ++b_)
            if ((int32_t)e[b_].y >= 0) push_raw(__uint_as_float(e[b_].x), (int32_t)e[b_].y);
    }
    for (int t = tid; t < spill; t += MG_THREADS)
        push_raw(cand_val[(size_t)row * cand_cap + t], cand_idx[(size_t)row * cand_cap + t]);
    __syncthreads();
    if (s_n > (uint32_t)MG_CAP) {               // block-uniform: more survivors than the list holds
        if (tid == 0) overflow_rows[atomicAdd(overflow_count, 1)] = (int32_t)(row_base + row);
        return;
    }
    uint32_t ncand = s_n;
    // entries that can still be among the K best unseen: those in or above the bin of the `need`-th largest key
    const uint32_t need = (uint32_t)K + (uint32_t)(deg < MG_CAP ? deg : MG_CAP);
    __shared__ uint32_t s_m;
    if (ncand > 2u * need && ncand > 128u) {
        __shared__ uint32_t hist[MG_BINS];
        __shared__ uint32_t s_lo, s_hi, s_bin, s_wv[MG_THREADS / 64];
        if (tid == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; s_m = 0u; }
        for (int b_ = tid; b_ < MG_BINS; b_ += MG_THREADS) hist[b_] = 0u;
        __syncthreads();
        uint32_t lo_k = 0xFFFFFFFFu, hi_k = 0u;
        for (uint32_t t = tid; t < ncand; t += MG_THREADS) {
            const uint32_t k32 = (uint32_t)(cand[t] >> 32);
            lo_k = k32 < lo_k ? k32 : lo_k; hi_k = k32 > hi_k ? k32 : hi_k;
        }
        atomicMin(&s_lo, lo_k); atomicMax(&s_hi, hi_k);
        __syncthreads();
        const uint32_t klo = s_lo;
        const uint32_t width = s_hi - klo;                         // largest offset, >= 0
        const uint32_t shift = width >= (uint32_t)MG_BINS ? (32u - (uint32_t)__clz(width)) - MG_BIN_BITS : 0u;
        for (uint32_t t = tid; t < ncand; t += MG_THREADS)
            atomicAdd(&hist[((uint32_t)(cand[t] >> 32) - klo) >> shift], 1u);
        __syncthreads();
        // bin of the need-th largest: suffix counts from the top bin (thread t owns MG_BINS/threads bins)
        constexpr int per = MG_BINS / MG_THREADS;
        uint32_t mine_b = 0;
        for (int q = 0; q < per; ++q) mine_b += hist[tid * per + q];
        uint32_t suf = mine_b;
        {
            const int ln = tid & 63;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_down(suf, o, 64); if (ln + o < 64) suf += x; }
            if (ln == 0) s_wv[tid >> 6] = suf;
            __syncthreads();
            for (int w_ = (tid >> 6) + 1; w_ < MG_THREADS / 64; ++w_) suf += s_wv[w_];
        }
        const uint32_t above = suf - mine_b;
        if (above < need && suf >= need) {
            uint32_t run = above;
            for (int q = per - 1; q >= 0; --q) {
                run += hist[tid * per + q];
                if (run >= need) { s_bin = (uint32_t)(tid * per + q); break; }
            }
        }
        __syncthreads();
        const uint32_t kbin = s_bin;
        // compact in place, a round of MG_THREADS elements at a time: everything a round reads is
        // in registers before anything is written, and writes only land below the read frontier
        for (uint32_t r0 = 0; r0 < ncand; r0 += MG_THREADS) {
            const uint32_t t = r0 + tid;
            unsigned long long e = 0ull;
            bool live = false;
            if (t < ncand) { e = cand[t]; live = (((uint32_t)(e >> 32) - klo) >> shift) >= kbin; }
            __syncthreads();
            if (live) cand[atomicAdd(&s_m, 1u)] = e;
        }
        __syncthreads();
        ncand = s_m;
    }
    // the entries that are left: id in the permuted table -> item id, seen items dropped (their entry becomes 0: below every real
    // key, it sorts last), the rest in the form the sort orders -- key descending, then item id ascending
    __syncthreads();
    if (tid == 0) s_m = 0u;
    __syncthreads();
    for (uint32_t t = tid; t < ncand; t += MG_THREADS) {
        const unsigned long long e = cand[t];
        const int32_t it = (int32_t)(((int64_t)(uint32_t)e * perm_a) % perm_n);
        bool is_seen = false;
        if (seen_lds) {
            int a = 0, z = deg;
            while (a < z) { const int m = (a + z) >> 1; if (seen[m] < it) a = m + 1; else z = m; }
            is_seen = a < deg && seen[a] == it;
        } else {
            int64_t a = lo, z = hi;             // a long row: binary search in the CSR itself
            while (a < z) { const int64_t m = (a + z) >> 1; if (indices[m] < it) a = m + 1; else z = m; }
            is_seen = a < hi && indices[a] == it;
        }
        if (RSX_ABL(32)) is_seen = false;       // (dev build only: the seen-item test of the merge skipped, tests/test_mutation.py)
        cand[t] = is_seen ? 0ull : ((e & 0xFFFFFFFF00000000ull) | (uint32_t)(0xFFFFFFFFu - (uint32_t)it));
        if (is_seen) atomicAdd(&s_m, 1u);
    }
    __syncthreads();
    const uint32_t n_valid = ncand - s_m;       // (>= K unless the whole catalog holds fewer unseen items)
    uint32_t n2 = 1;
    while (n2 < ncand) n2 <<= 1;
    for (uint32_t t = ncand + tid; t < n2; t += MG_THREADS) cand[t] = 0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < n2 / 2; t += MG_THREADS) {
                const uint32_t l = 2 * t - (t & (stride - 1));
                const uint32_t h = l + stride;
                const bool desc = ((l & size) == 0);
                const unsigned long long a = cand[l], b = cand[h];
                if ((a < b) == desc) { cand[l] = b; cand[h] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < K; t += MG_THREADS) {
        const unsigned long long e = cand[t];
        out_idx[(size_t)row * K + t] = (t < (int)n_valid) ? (int32_t)(0xFFFFFFFFu - (uint32_t)e) : -1;
        if (out_val) out_val[(size_t)row * K + t] = (t < (int)n_valid) ? key2f((uint32_t)(e >> 32)) : -INFINITY;
    }
}

constexpr int kMaxLanes = 4;   // g_rsx_score_lanes (rsx_set_option "score_lanes"): passes of the fused path in flight

template <bool FILTER>
int launch_score(const float *P, const int32_t *users, int64_t rows, const float *Q, int64_t cols,
                 int64_t item_stride, int d, float *out, const float *tau, float *cand_val,
                 int32_t *cand_idx, int32_t *cand_cnt, int cand_cap, uint2 *slots, hipStream_t st, int64_t item_base = 0)
{
    const int64_t n_item_tiles = (cols + BN - 1) / BN;
    const int64_t n_row_tiles = (rows + BM - 1) / BM, n_tiles = n_item_tiles * n_row_tiles;
    dim3 grid((unsigned)((n_item_tiles + RSX_SCORE_TILES_PER_WG - 1) / RSX_SCORE_TILES_PER_WG), (unsigned)n_row_tiles);
    // the LDS-DMA form addresses item rows by 32-bit byte offsets: item tables below 4 GB (every BASELINE catalog: 1M x 128 = 512 MB)
    const bool glds = RSX_SCORE_GLDS != 0 && cols * item_stride * (int64_t)d * 4 < (1ll << 32);
#define RSX_SCORE_LAUNCH(D_) do { if (glds) hipLaunchKernelGGL((score_tile_glds_kernel<D_, FILTER>), grid, dim3(256), 0, st, P, users, rows, Q, cols, item_stride, out, tau, cand_val, cand_idx, cand_cnt, cand_cap, slots, item_base, n_item_tiles, n_tiles); \
                                  else hipLaunchKernelGGL((score_tile_kernel<D_, FILTER>), grid, dim3(256), 0, st, P, users, rows, Q, cols, item_stride, out, tau, cand_val, cand_idx, cand_cnt, cand_cap, slots, item_base, n_item_tiles, n_tiles); } while (0)
    switch (d) {
    case 32: RSX_SCORE_LAUNCH(32); break;
    case 64: RSX_SCORE_LAUNCH(64); break;
    case 128: RSX_SCORE_LAUNCH(128); break;
    default: RSX_SCORE_LAUNCH(256); break;
    }
#undef RSX_SCORE_LAUNCH
    return 0;
}

constexpr int64_t kRowTile = 1024;   // rows scored per pass of rsx_score_topk (evaluator.py:11 batch)

// side streams of the fused path, one set per device, created on first use under a lock
struct LanePool {
    std::mutex busy;
    hipStream_t side[kMaxLanes - 1];
    hipEvent_t fork, join[kMaxLanes - 1];
    bool ready = false;
};

LanePool *lane_pool()
{
    static LanePool pools[64];
    static std::mutex create;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    LanePool &p = pools[dev];
    std::lock_guard<std::mutex> g(create);
    if (!p.ready) {
        bool ok = hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) == hipSuccess;
        for (int l = 0; ok && l < kMaxLanes - 1; ++l)
            ok = hipStreamCreateWithFlags(&p.side[l], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&p.join[l], hipEventDisableTiming) == hipSuccess;
        if (!ok) return nullptr;
        p.ready = true;
    }
    return &p;
}

}  // namespace

#ifdef RSX_ABLATE
// dev build only (librsx_dev.so).  2 / 4: write switches of tools/ablate_score.py; planted errors of tests/test_mutation.py:
//   8 the last K chunk of the product dropped, 16 the filter threshold lifted, 32 the merge's seen-item test skipped,
//   64 the dense row Top-K hands out the (K+1)-th candidate in the K-th place
RSX_API int rsx_debug_set_score_ablation(int mask)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_rsx_ablate), &mask, sizeof(int)) == hipSuccess ? RSX_OK : RSX_E_HIP;
}
// dev build only: a threshold per row of the NEXT rsx_score_topk calls (device array over the call's rows, or NULL) that replaces the
// sample's tau in the filtered product -- any lower bound of the row's K-th score gives the same result.  It prices what a tighter
// threshold (e.g. one refreshed inside a pass) could save before building it: tools/exp_tau_bound.py.
static const float *g_tau_override = nullptr;
RSX_API int rsx_debug_set_tau_override(const float *tau_dev) { g_tau_override = tau_dev; return RSX_OK; }
#endif

RSX_API int rsx_score(const float *P, const int32_t *user_ids_dev, int64_t num_rows, const float *Q,
                      int64_t num_items, int d, const int64_t *mask_indptr_dev,
                      const int32_t *mask_indices_dev, float *scores_out, rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && user_ids_dev && scores_out, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(num_rows >= 0 && num_items > 0, "bad shape");
    RSX_CHECK_ARG((mask_indptr_dev == nullptr) == (mask_indices_dev == nullptr), "mask needs both CSR arrays");
    if (num_rows == 0) return RSX_OK;
    hipStream_t st = (hipStream_t)stream;
    for (int64_t r0 = 0; r0 < num_rows; r0 += 65535 * (int64_t)BM) {   // gridDim.y limit
        const int64_t nr = (num_rows - r0 < 65535 * (int64_t)BM) ? num_rows - r0 : 65535 * (int64_t)BM;
        launch_score<false>(P, user_ids_dev + r0, nr, Q, num_items, 1, d, scores_out + (size_t)r0 * num_items,
                            nullptr, nullptr, nullptr, nullptr, 0, nullptr, st);
    }
    if (mask_indptr_dev) {
        for (int64_t r0 = 0; r0 < num_rows; r0 += (1ll << 30)) {
            const int64_t nr = (num_rows - r0 < (1ll << 30)) ? num_rows - r0 : (1ll << 30);
            hipLaunchKernelGGL(mask_seen_kernel, dim3((unsigned)nr), dim3(256), 0, st,
                               scores_out + (size_t)r0 * num_items, user_ids_dev + r0, nr, num_items,
                               mask_indptr_dev, mask_indices_dev);
        }
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_topk(const float *scores_dev, int64_t num_rows, int64_t num_items, int K,
                     int32_t *topk_idx_out, float *topk_val_out, rsx_stream_t stream)
{
    RSX_CHECK_ARG(scores_dev && topk_idx_out, "null pointer");
    RSX_CHECK_ARG(K >= 1 && K <= TK_CAP / 2 && K <= num_items, "K must be in [1, min(1024, num_items)]");
    RSX_CHECK_ARG(num_rows >= 0 && num_rows < (1ll << 31) && num_items < (1ll << 32) - 1, "bad shape");
    if (num_rows == 0) return RSX_OK;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)num_rows), dim3(TK_THREADS), 0,
                       (hipStream_t)stream, scores_dev, num_items, K, topk_idx_out, topk_val_out);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

namespace {

constexpr int64_t kSampleCols = RSX_SAMPLE_COLS;      // sample size of the fused path
#ifndef RSX_FUSED_ROWS
#define RSX_FUSED_ROWS 8192
#endif
constexpr int64_t kFusedRows = RSX_FUSED_ROWS;       // rows per pass of the fused path: one launch of ~50K workgroups (development A/B: 4096 / 16384)
                                           // (a 1024-row pass is 8.15 rounds of 768 resident workgroups:
                                           //  11 % of the time is the ragged last round, plus 7 launches)
constexpr int64_t kFusedMinItems = 4 * kSampleCols;
constexpr int64_t kFallbackRows = 64;      // dense re-do granularity for overflowed rows
constexpr int kSpillCap = 1024;            // per-row spill list of the filtered product

int64_t a256(int64_t x) { return (x + 255) / 256 * 256; }

bool use_fused(int64_t num_items, int K) { return num_items >= kFusedMinItems && K <= 512; }

struct FusedWs {
    float *sample;      // [rows x n_s]
    float *tau;         // [rows]
    float *topv;        // [rows x K]
    int32_t *topi;      // [rows x K]
    float *cval;        // [rows x kSpillCap]   spill lists (cells with more than kSlots survivors)
    int32_t *cidx;      // [rows x kSpillCap]
    int32_t *ccnt;      // [rows]
    uint2 *slots;       // [rows x 64-item strips x kSlots] {score bits, item}; 0xFF-filled per pass
    int32_t *ovf_rows;  // [rows]
    int32_t *ovf_cnt;   // [1]
    int32_t *fb_users;  // [kFallbackRows]
    float *fb_scores;   // [kFallbackRows x num_items]
    int64_t bytes;
};

FusedWs carve(void *ws, int64_t tile_rows, int64_t all_rows, int64_t num_items, int K)
{
    FusedWs w;
    int64_t off = 0;
    auto take = [&](int64_t n) { char *q = (char *)ws + off; off += a256(n); return q; };
    w.sample = (float *)take(tile_rows * kSampleCols * 4);
    w.tau = (float *)take(tile_rows * 4);
    w.topv = (float *)take(tile_rows * K * 4);
    w.topi = (int32_t *)take(tile_rows * K * 4);
    w.cval = (float *)take(tile_rows * kSpillCap * 4);
    w.cidx = (int32_t *)take(tile_rows * kSpillCap * 4);
    w.ccnt = (int32_t *)take(tile_rows * 4);
    const int64_t n_it = 2 * ((num_items + BN - 1) / BN), rows_pad = (tile_rows + BM - 1) / BM * BM;
    w.slots = (uint2 *)take(n_it * rows_pad * kSlots * 8);
    w.ovf_rows = (int32_t *)take(all_rows * 4);
    w.ovf_cnt = (int32_t *)take(4);
    w.fb_users = (int32_t *)take(kFallbackRows * 4);
    w.fb_scores = (float *)take(kFallbackRows * num_items * 4);
    w.bytes = off;
    return w;
}

}  // namespace

RSX_API int64_t rsx_score_topk_workspace_d(int64_t num_rows, int64_t num_items, int d)
{
    if (num_rows < 0 || num_items <= 0 || !rsx_dim_ok(d)) return RSX_E_INVALID;
    const int64_t rows = num_rows < kRowTile ? num_rows : kRowTile;
    const int64_t dense = rows * num_items * 4;
    if (num_items < kFusedMinItems) return dense;
    const int64_t frows = num_rows < kFusedRows ? num_rows : kFusedRows;
    const int64_t passes = (num_rows + kFusedRows - 1) / kFusedRows;
    const int64_t lanes = passes < kMaxLanes ? passes : kMaxLanes;   // passes in flight (one stream each)
    const int64_t fused = lanes * carve(nullptr, frows, num_rows, num_items, 512).bytes + a256(num_items * (int64_t)d * 4);   // + Qp, the permuted item table
    return fused > dense ? fused : dense;   // (K > 512 still takes the dense path)
}

// the bound for any row width (d = 256): what a caller that does not know d yet reserves
RSX_API int64_t rsx_score_topk_workspace(int64_t num_rows, int64_t num_items)
{
    return rsx_score_topk_workspace_d(num_rows, num_items, 256);
}

RSX_API int rsx_score_topk(const float *P, const int32_t *user_ids_dev, int64_t num_rows,
                           const float *Q, int64_t num_items, int d, const int64_t *mask_indptr_dev,
                           const int32_t *mask_indices_dev, int K, int32_t *topk_idx_out,
                           float *topk_val_out, void *ws, int64_t ws_bytes, rsx_stream_t stream)
{
    RSX_CHECK_ARG(topk_idx_out != nullptr, "null output");
    const int64_t need = rsx_score_topk_workspace_d(num_rows, num_items, rsx_dim_ok(d) ? d : 256);
    if (need < 0) { rsx_set_error("rsx_score_topk: bad shape"); return RSX_E_INVALID; }
    if (num_rows == 0) return RSX_OK;
    if (ws == nullptr || ws_bytes < need) {
        rsx_set_error("rsx_score_topk: workspace of %lld bytes required, got %lld", (long long)need,
                      (long long)ws_bytes);
        return RSX_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!use_fused(num_items, K)) {            // small catalogs: dense tile + row top-k
        float *tile = (float *)ws;
        for (int64_t r0 = 0; r0 < num_rows; r0 += kRowTile) {
            const int64_t nr = (num_rows - r0 < kRowTile) ? num_rows - r0 : kRowTile;
            int rc = rsx_score(P, user_ids_dev + r0, nr, Q, num_items, d, mask_indptr_dev, mask_indices_dev,
                               tile, stream);
            if (rc != RSX_OK) return rc;
            rc = rsx_topk(tile, nr, num_items, K, topk_idx_out + (size_t)r0 * K,
                          topk_val_out ? topk_val_out + (size_t)r0 * K : nullptr, stream);
            if (rc != RSX_OK) return rc;
        }
        return RSX_OK;
    }
    RSX_CHECK_ARG(P && Q && user_ids_dev, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(K >= 1 && K <= num_items, "K must be in [1, num_items]");
    RSX_CHECK_ARG((mask_indptr_dev == nullptr) == (mask_indices_dev == nullptr), "mask needs both CSR arrays");
    const int64_t tile_rows = num_rows < kFusedRows ? num_rows : kFusedRows;
    const int64_t n_tiles = (num_rows + kFusedRows - 1) / kFusedRows;
    // Several passes in flight, one HIP stream each: the selection kernels of one pass (sample
    // top-K, merge; memory / LDS bound) overlap the matrix-core product of the others.
    const int want_lanes = g_rsx_score_lanes < 1 ? 1 : (g_rsx_score_lanes > kMaxLanes ? kMaxLanes : g_rsx_score_lanes);
    const int n_lanes = (int)(n_tiles < want_lanes ? n_tiles : want_lanes);
    FusedWs lane_ws[kMaxLanes];
    for (int l = 0; l < n_lanes; ++l)
        lane_ws[l] = carve((char *)ws + (l ? (size_t)l * lane_ws[0].bytes : 0), tile_rows, num_rows, num_items, K);
    FusedWs &w = lane_ws[0];                   // overflow list and dense re-do buffers are lane 0's
    // the permuted copy of the item table (after the lanes' regions) and its multiplier: the smallest a >= I / sample
    // coprime to I; a^-1 mod I by the extended Euclidean algorithm
    float *Qp = (float *)((char *)ws + (size_t)n_lanes * lane_ws[0].bytes);
    int64_t perm_a = num_items / kSampleCols, perm_inv = 1;
    for (;; ++perm_a) {
        int64_t x = perm_a, y = num_items;
        while (y) { const int64_t t = x % y; x = y; y = t; }
        if (x == 1) break;
    }
    {
        int64_t r0_ = num_items, r1 = perm_a % num_items, t0 = 0, t1 = 1;
        while (r1) { const int64_t q = r0_ / r1, r2 = r0_ - q * r1, t2 = t0 - q * t1; r0_ = r1; r1 = r2; t0 = t1; t1 = t2; }
        perm_inv = ((t0 % num_items) + num_items) % num_items;
    }
    // side streams and fork/join events belong to the CURRENT device; one caller at a time per device
    // uses them (the lock is held until every launch of this call has been queued)
    LanePool *pool = lane_pool();
    if (pool == nullptr) { rsx_set_error("rsx_score_topk: could not create the side streams"); return RSX_E_HIP; }
    std::lock_guard<std::mutex> guard(pool->busy);
    hipStream_t *side_stream = pool->side;
    hipEvent_t ev_fork = pool->fork, *ev_join = pool->join;
    hipStream_t lane_st[kMaxLanes] = {st, st, st, st};
    for (int l = 1; l < n_lanes; ++l) lane_st[l] = side_stream[l - 1];
    (void)hipMemsetAsync(w.ovf_cnt, 0, 4, st);
    hipLaunchKernelGGL(permute_items_kernel, dim3((unsigned)(rsx_num_cus() * 8)), dim3(256), 0, st, (const float4 *)Q,
                       (float4 *)Qp, num_items, d / 4, perm_a);
    if (n_lanes > 1) {                          // (the lanes start after the permuted table is complete)
        (void)hipEventRecord(ev_fork, st);
        for (int l = 1; l < n_lanes; ++l) (void)hipStreamWaitEvent(lane_st[l], ev_fork, 0);
    }
    for (int64_t ti = 0; ti < n_tiles; ++ti) {
        const int64_t r0 = ti * kFusedRows;
        const int64_t nr = (num_rows - r0 < kFusedRows) ? num_rows - r0 : kFusedRows;
        const int32_t *users = user_ids_dev + r0;
        FusedWs &lw = lane_ws[ti % n_lanes];
        hipStream_t ls = lane_st[ti % n_lanes];
        // 1. the sample = the FIRST kSampleCols rows of the permuted table, scored densely; tau = the K-th best of its unseen items, and those
        //    K (or more, on ties) go to the row's candidate list: the sample is not scored a second time
        launch_score<false>(P, users, nr, Qp, kSampleCols, 1, d, lw.sample, nullptr, nullptr, nullptr, nullptr, 0,
                            nullptr, ls);
        static_assert(kSampleCols == ST_COLS, "sample_tau_kernel is laid out for the sample size");
        hipLaunchKernelGGL(sample_tau_kernel, dim3((unsigned)nr), dim3(TK_THREADS), 0, ls, lw.sample, users, nr, perm_inv, num_items,
                           mask_indptr_dev, mask_indices_dev, K, lw.tau, lw.ccnt, lw.cval, lw.cidx, kSpillCap);
#ifdef RSX_ABLATE
        if (g_tau_override != nullptr) (void)hipMemcpyAsync(lw.tau, g_tau_override + r0, (size_t)nr * 4, hipMemcpyDeviceToDevice, ls);
#endif
        // 2. the rest of the catalog with the FILTER epilogue (item ids offset by the sample)
        const int64_t rest = num_items - kSampleCols;
        const int64_t n_it = 2 * ((rest + BN - 1) / BN), rows_pad = (nr + BM - 1) / BM * BM;
        if (n_it * rows_pad * kSlots * 8 >= (1ll << 32)) {      // the filter epilogue addresses its slots with 32-bit byte offsets
            rsx_set_error("rsx_score_topk: catalogs above ~2 million items are not supported by the fused path (slot array of a pass >= 4 GB)");
            return RSX_E_INVALID;
        }
        (void)hipMemsetAsync(lw.slots, 0xFF, (size_t)(n_it * rows_pad) * kSlots * 8, ls);
        launch_score<true>(P, users, nr, Qp + (size_t)kSampleCols * d, rest, 1, d, nullptr, lw.tau, lw.cval, lw.cidx, lw.ccnt,
                           kSpillCap, lw.slots, ls, kSampleCols);
        hipLaunchKernelGGL(merge_candidates_kernel, dim3((unsigned)nr), dim3(MG_THREADS), 0, ls, lw.slots,
                           n_it, rows_pad, lw.cval, lw.cidx, lw.ccnt, kSpillCap, users, mask_indptr_dev,
                           mask_indices_dev, K,
                           topk_idx_out + (size_t)r0 * K, topk_val_out ? topk_val_out + (size_t)r0 * K : nullptr,
                           r0, w.ovf_rows, w.ovf_cnt, perm_a, num_items);
        RSX_CHECK_LAUNCH();
    }
    for (int l = 1; l < n_lanes; ++l) {
        (void)hipEventRecord(ev_join[l - 1], lane_st[l]);
        (void)hipStreamWaitEvent(st, ev_join[l - 1], 0);
    }
    // rows whose candidate list overflowed (massive exact ties, or fewer than K unmasked sample
    // items) are re-done through the dense path.  This is the one place the call waits for the
    // stream: it has to read the overflow count.
    int32_t h_cnt = 0;
    hipError_t e = hipMemcpyAsync(&h_cnt, w.ovf_cnt, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { rsx_set_error("rsx_score_topk: %s", hipGetErrorString(e)); return RSX_E_HIP; }
    if (h_cnt > 0) {
        std::vector<int32_t> h_rows((size_t)h_cnt), h_users((size_t)num_rows);
        (void)hipMemcpy(h_rows.data(), w.ovf_rows, sizeof(int32_t) * (size_t)h_cnt, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h_users.data(), user_ids_dev, sizeof(int32_t) * (size_t)num_rows, hipMemcpyDeviceToHost);
        for (int64_t o = 0; o < h_cnt; o += kFallbackRows) {
            const int64_t nb = (h_cnt - o < kFallbackRows) ? h_cnt - o : kFallbackRows;
            int32_t fb[kFallbackRows];
            for (int64_t q = 0; q < nb; ++q) fb[q] = h_users[(size_t)h_rows[(size_t)(o + q)]];
            (void)hipMemcpy(w.fb_users, fb, sizeof(int32_t) * (size_t)nb, hipMemcpyHostToDevice);
            int rc = rsx_score(P, w.fb_users, nb, Q, num_items, d, mask_indptr_dev, mask_indices_dev, w.fb_scores, stream);
            if (rc != RSX_OK) return rc;
            for (int64_t q = 0; q < nb; ++q) {
                const int64_t grow = h_rows[(size_t)(o + q)];
                rc = rsx_topk(w.fb_scores + (size_t)q * num_items, 1, num_items, K, topk_idx_out + (size_t)grow * K,
                              topk_val_out ? topk_val_out + (size_t)grow * K : nullptr, stream);
                if (rc != RSX_OK) return rc;
            }
            (void)hipStreamSynchronize(st);     // fb_users / fb_scores are reused by the next group
        }
    }
    return RSX_OK;
}
