// rsx_sample.hip -- on-device BPR triplet sampler for gfx950 (MI355X).
//
// Replaces the reference's host-side PairwiseGenerator (data/generators.py:151-224:
// per-user numpy sampling + permutation + H2D copy per batch).  Semantics (documented
// divergences from the reference's quirks are in DESIGN.md 4.3):
//   user   position (epoch_pos + b) of a keyed pseudo-random permutation of the users
//   pos i  uniform over the user's CSR row
//   neg j  uniform over candidate items, rejected while j is in that row
// The batch is a SET (the step is batch-synchronous), so its order is free.  Two layouts:
//   plain   one kernel, triplet b sits at batch position b
//   sorted  (RSX_SAMPLE_SORT_POS) the (i, u) pairs are ordered by positive item, then
//           negatives are drawn per ordered position from a keyed item block.
//           Equal positives become contiguous, so the step kernel sums their gradient
//           in registers, and all negatives of an item block belong to one wavefront.
//           With the item CDF (rsx_bpr_build_item_cdf) the order comes from this file's
//           bucket kernels; without it from rocPRIM's device radix sort (a plain library
//           primitive, like a library GEMM).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "rsx_common.h"

namespace {

constexpr int kBlock = 256;
#ifdef RSX_ABLATE
__device__ uint32_t *g_mock_neg = nullptr;       // [kPiece] scratch of the redesign mock (development build, see rsx_debug_set_sample_ablation)
#endif
constexpr uint64_t kSigStartMask = (1ull << 40) - 1, kSigLenClip = (1ull << 24) - 1;

// rocPRIM's default policy (merge sort up to 1M pairs, ~195 us for 1M on MI355X) measured
// faster here than forcing Onesweep (MergeSortLimit = 0: ~415 us for 1M 17-bit keys)
using sort_config = rocprim::default_config;

__device__ __forceinline__ uint32_t rng_seed(uint64_t seed, uint64_t step, uint64_t b, uint64_t salt)
{
    uint32_t s = (uint32_t)splitmix64(seed ^ (step * 0x9E3779B97F4A7C15ull) ^ (b * 0xBF58476D1CE4E5B9ull) ^ salt);
    return s | (s == 0);
}

// hb < 0: the batch holds every user exactly once (batch == num_users), so no permutation is
// needed -- a batch is a set -- and walking the users in id order makes the indptr / row reads
// of the sampling pass coalesced instead of one random sector per user.
__device__ __forceinline__ uint32_t user_at(int64_t gpos, int64_t U, int hb, uint64_t seed)
{
    const uint64_t epoch = (uint64_t)(gpos / U);
    const uint32_t pos = (uint32_t)(gpos % U);
    if (hb < 0) return pos;
    return feistel_perm(pos, (uint32_t)U, hb, splitmix64(seed ^ (epoch * 0xD1B54A32D192ED03ull)));
}

__device__ __forceinline__ bool row_has(const int32_t *__restrict__ indices, int64_t lo, int64_t hi, int32_t x)
{
    int64_t a = lo, z = hi;     // binary search in the sorted row
    while (a < z) {
        const int64_t m = (a + z) >> 1;
        if (indices[m] < x) a = m + 1; else z = m;
    }
    return a < hi && indices[a] == x;
}

// negative for batch position p: uniform in the candidate range, not in the user's row
__device__ __forceinline__ int sig_bit(int64_t block) { return (int)(splitmix64((uint64_t)block) & 63u); }

__device__ __forceinline__ void neg_range(int64_t I, int64_t p, int64_t B, int neg_block, uint64_t neg_key,
                                          int64_t &neg_lo, int64_t &neg_n)
{
    neg_lo = 0; neg_n = I;
    if (neg_block > 0) {
        const int64_t nblocks = ceil_div64(I, neg_block);
        const int64_t w = ((p * I) / B) / neg_block;
        neg_lo = neg_block_of(w, nblocks, neg_key) * neg_block;
        neg_n = (neg_lo + neg_block <= I) ? neg_block : I - neg_lo;
    }
}

__device__ __forceinline__ int32_t draw_negative(const int32_t *__restrict__ indices, int64_t lo, int64_t hi,
                                                 int64_t I, int64_t p, int64_t B, int neg_block,
                                                 uint64_t neg_key, uint32_t &s)
{
    int64_t neg_lo, neg_n;
    neg_range(I, p, B, neg_block, neg_key, neg_lo, neg_n);
    for (int tries = 0;; ++tries) {
        if (tries == 64) { neg_lo = 0; neg_n = I; }   // the user owns (nearly) the whole block
        const int32_t nj = (int32_t)(neg_lo + (int64_t)(((uint64_t)xorshift32(s) * (uint64_t)neg_n) >> 32));
        if (!row_has(indices, lo, hi, nj)) return nj;
    }
}

// plain layout: everything for position b in one pass
__global__ __launch_bounds__(kBlock) void bpr_sample_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t U, int64_t I,
    int64_t B, uint64_t seed, uint64_t step, int64_t epoch_pos, int hb, int neg_block, uint64_t neg_key,
    int32_t *__restrict__ u_out, int32_t *__restrict__ i_out, int32_t *__restrict__ j_out)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        const uint32_t u = user_at(epoch_pos + b, U, hb, seed);
        uint32_t s = rng_seed(seed, step, (uint64_t)b, 0);
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        const uint32_t deg = (uint32_t)(hi - lo);
        int32_t pi = -1, nj = -1;
        if (deg > 0 && (int64_t)deg < I) {
            pi = indices[lo + (int64_t)(((uint64_t)xorshift32(s) * deg) >> 32)];
            nj = draw_negative(indices, lo, hi, I, b, B, neg_block, neg_key, s);
        }
        u_out[b] = (int32_t)u; i_out[b] = pi; j_out[b] = nj;
    }
}

// sorted layout, pass 1: key = positive item (num_items = "no positive", sorts last), value = user
__global__ __launch_bounds__(kBlock) void sample_ui_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t U, int64_t I,
    int64_t B, uint64_t seed, uint64_t step, int64_t epoch_pos, int hb,
    uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        const uint32_t u = user_at(epoch_pos + b, U, hb, seed);
        uint32_t s = rng_seed(seed, step, (uint64_t)b, 0);
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        const uint32_t deg = (uint32_t)(hi - lo);
        uint32_t key = (uint32_t)I;
        if (deg > 0 && (int64_t)deg < I) key = (uint32_t)indices[lo + (int64_t)(((uint64_t)xorshift32(s) * deg) >> 32)];
        keys[b] = key; vals[b] = u;
    }
}

// user_sig[2u] = OR over the user's positives of (1 << sig_bit(item / neg_block)); user_sig[2u+1] =
// row start | row length << 40 (length clipped to 2^24-1: "look it up in indptr"); static per CSR
__global__ __launch_bounds__(kBlock) void build_signature_kernel(const int64_t *__restrict__ indptr,
                                                                 const int32_t *__restrict__ indices, int64_t U,
                                                                 int neg_block, uint64_t *__restrict__ sig)
{
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < U; u += (int64_t)gridDim.x * kBlock) {
        uint64_t m = 0ull;
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        for (int64_t q = lo; q < hi; ++q) m |= 1ull << sig_bit(indices[q] / neg_block);
        const uint64_t len = (uint64_t)(hi - lo) < kSigLenClip ? (uint64_t)(hi - lo) : kSigLenClip;
        sig[2 * u] = m;
        sig[2 * u + 1] = ((uint64_t)lo & kSigStartMask) | (len << 40);
    }
}

// sorted layout, pass 2: negatives per SORTED position p
__global__ __launch_bounds__(kBlock) void sample_neg_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t I, int64_t B,
    uint64_t seed, uint64_t step, int neg_block, uint64_t neg_key,
    const uint32_t *__restrict__ keys_sorted, const uint32_t *__restrict__ vals_sorted,
    const uint64_t *__restrict__ user_sig,
    int32_t *__restrict__ u_out, int32_t *__restrict__ i_out, int32_t *__restrict__ j_out)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < B; p += (int64_t)gridDim.x * kBlock) {
        const uint32_t u = vals_sorted[p];
        const uint32_t key = keys_sorted[p];
        int32_t pi = -1, nj = -1;
        if ((int64_t)key < I) {
            uint32_t s = rng_seed(seed, step, (uint64_t)p, 0x5bd1e995ull);
            pi = (int32_t)key;
            bool done = false;
            if (user_sig != nullptr && neg_block > 0) {      // clear signature bit: the whole block is negative
                int64_t neg_lo, neg_n;
                neg_range(I, p, B, neg_block, neg_key, neg_lo, neg_n);
                if (((user_sig[2 * (size_t)u] >> sig_bit(neg_lo / neg_block)) & 1ull) == 0ull) {
                    nj = (int32_t)(neg_lo + (int64_t)(((uint64_t)xorshift32(s) * (uint64_t)neg_n) >> 32));
                    done = true;
                }
            }
            if (!done) nj = draw_negative(indices, indptr[u], indptr[u + 1], I, p, B, neg_block, neg_key, s);
        }
        u_out[p] = (int32_t)u; i_out[p] = pi; j_out[p] = nj;
    }
}


// ---- sorted layout without a device-wide sort (item_cdf given) --------------------------------
// A device radix sort is a chain of small dependent kernels with decoupled look-back; run beside
// the step kernel (which fills every wave slot) it stretched from 79 us to ~400 us and became the
// critical path of the training step.  The sampling distribution over positive items is static per
// CSR, so the batch is cut into BALANCED item-range buckets from that distribution's CDF instead:
//   bucket_chunk_kernel   one workgroup per 4096 positions: sample (u, i), bucket = floor(NB * t),
//                         t a point of item i's CDF interval chosen by a hash of u (a popular item
//                         spreads over several buckets in proportion), LDS counting sort of the
//                         chunk by bucket, chunk + its per-bucket (offset, count) row written out
//   bucket_sort_kernel    one workgroup per bucket (~832 pairs): gather the bucket's slices of all
//                         chunks into LDS, bitonic sort by (item, user), draw the negatives per
//                         final position, write u/i/j
// Buckets are monotone in the item, so the concatenation is ordered by positive item; the order
// is a pure function of the sampled set (no atomics decide a position).
#ifndef RSX_CHUNK_POS
#define RSX_CHUNK_POS 4096
#endif
constexpr int kChunk = RSX_CHUNK_POS;     // positions per bucket_chunk workgroup (development A/B: 2048 / 1024 -- more, shorter workgroups)
// workgroup of the bucketing pass.  It runs beside the step kernel, whose wavefronts hold 480 of a SIMD's 512 VGPRs: a
// 1024-thread workgroup needs four wavefronts (4 x 48 VGPRs) on EVERY SIMD of one CU at once, a 256-thread one a single one
// (122 VGPRs: sixteen positions per thread in lockstep).  Same box, 300 steps at the headline shape, us per step: with item
// blocks of 6 (long step wavefronts) 340 / 331 / 355 at 1024 / 512 / 256 threads; with blocks of 2-3 (short ones, what
// sharded.py:pick_neg_block now picks there) 345 / 336 / 327 -- profiles/r03_exp_sampler_placement.txt
#ifndef RSX_CHUNK_THREADS
#define RSX_CHUNK_THREADS 256
#endif
constexpr int kChunkThreads = RSX_CHUNK_THREADS;
constexpr int kPerThread = kChunk / kChunkThreads;
#ifndef RSX_BUCKET_MEAN
#define RSX_BUCKET_MEAN 832
#endif
constexpr int kBucketMean = RSX_BUCKET_MEAN;          // expected pairs per bucket: 1024 - 6.6 sigma (development A/B: 416 / 1664)
constexpr int kSortCap = 2048;            // pairs a bucket may hold and still be sorted in LDS
constexpr int64_t kPiece = 1ll << 21;     // positions bucketed per pass (bounds LDS bins and workspace)
constexpr int kMaxBuckets = (int)(kPiece / kBucketMean) + 3 + RSX_MAX_CHUNKS;
constexpr int kMaxChunks = (int)(kPiece / kChunk);        // 512
constexpr int kTotalStride = 32;          // bucket totals sit one per 128-B line (same-line atomics serialise)
constexpr int kRangeCap = 256;            // negative-block ranges a bucket's positions may span and still use the LDS table
static_assert((kMaxBuckets + 4) * 4 + kChunk * 8 <= 64 * 1024, "bucket_chunk_kernel's LDS");

// g_rsx_sort_cap (rsx_set_option "sample_sort_cap"; 0 = kSortCap) lowers the LDS sort capacity so that
// tests can exercise the out-of-LDS path
static inline int lds_sort_cap() { return (g_rsx_sort_cap >= 1 && g_rsx_sort_cap <= kSortCap) ? g_rsx_sort_cap : kSortCap; }

// adjacent table entries fetched with one load (dword / qword alignment is enough for global loads)
struct __attribute__((packed, aligned(4))) U32Pair { uint32_t a, b; };
struct __attribute__((packed, aligned(8))) I64Pair { int64_t a, b; };

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// item chunks (include/rsx.h): C <= 1 is the plain layout.  NBc = buckets per range (the buckets never straddle a range).
struct ChunkArgs {
    int C, NBc;
    ChunkGeom g;
    int64_t *chunk_pos_out;      // [C + 1]
    bool whole;                  // negatives uniform over the REAL items of the position's range (neg_block = 0: no blocks)
};

// negative range of the position `rel` (relative to its range's first position; the range holds nc live positions) inside
// range ch: block pi_ch(w) of the range, w = floor(floor(rel * Ic / nc) / c); lo = first item id, n = REAL items in the block
__device__ __forceinline__ void neg_range_chunk(const ChunkGeom &g, int ch, int64_t rel, int64_t nc, uint64_t neg_key,
                                                int64_t &neg_lo, int64_t &neg_n)
{
    const int64_t w = ((rel * g.Ic) / nc) / g.c;
    const int64_t blk = neg_block_of(w, g.nbc, chunk_key(neg_key, ch));
    neg_lo = (int64_t)ch * g.Ic + blk * g.c;
    const int64_t left = g.real(ch) - blk * g.c;
    neg_n = left <= 0 ? 0 : (left < g.c ? left : g.c);
}

__global__ __launch_bounds__(kChunkThreads) void bucket_chunk_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, const uint32_t *__restrict__ cdf,
    int64_t U, int64_t I, int64_t piece_lo, int64_t n, uint64_t seed, uint64_t step, int64_t epoch_pos, int hb,
    int nbm, int nblk, uint2 *__restrict__ pairs, uint32_t *__restrict__ table, uint32_t *__restrict__ totals, ChunkArgs ca)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int NB = nbm + 1;
    uint32_t *hist = smem;                                    // [NB] counts, then exclusive offsets
    uint2 *stage = (uint2 *)(smem + ((NB + 3) & ~3));         // [kChunk]
    __shared__ uint32_t wave_sum[kChunkThreads / 64];
    // item chunks: the CDF at the ranges' borders and NBc / (mass of the range), once per workgroup (they used to be fetched and
    // divided -- a 64-bit division -- per position, behind the position's other loads: 42 -> 92 us stand-alone at C = 3)
    __shared__ uint32_t rng_lo[RSX_MAX_CHUNKS + 1];
    __shared__ double rng_scale[RSX_MAX_CHUNKS];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    for (int q = tid; q < NB; q += kChunkThreads) hist[q] = 0u;
    if (ca.C > 1 && tid <= ca.C) {
        const uint32_t lo = cdf[(int64_t)tid * ca.g.Ic];
        rng_lo[tid] = lo;
        if (tid < ca.C) {
            const uint32_t hi = cdf[(int64_t)(tid + 1) * ca.g.Ic];
            rng_scale[tid] = hi > lo ? (double)ca.NBc / (double)(hi - lo) : 0.0;
        }
    }
    __syncthreads();
    // the kPerThread positions of a thread advance in lockstep, one dependent load level at a
    // time, so that their global loads are in flight together (the pass is latency-bound)
    uint32_t eu[kPerThread], ei[kPerThread], eb[kPerThread];  // user, item, bucket << 13 | rank in bucket
    int64_t rlo[kPerThread], rhi[kPerThread];
    bool ok[kPerThread];
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
        const int64_t loc = (int64_t)blk * kChunk + e * kChunkThreads + tid;
        ok[e] = loc < n;
        eu[e] = ok[e] ? user_at(epoch_pos + piece_lo + loc, U, hb, seed) : 0u;
        const I64Pair rb = *reinterpret_cast<const I64Pair *>(indptr + eu[e]);
        rlo[e] = rb.a; rhi[e] = rb.b;
    }
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
        const int64_t b = piece_lo + (int64_t)blk * kChunk + e * kChunkThreads + tid;
        uint32_t s = rng_seed(seed, step, (uint64_t)b, 0);
        const uint32_t deg = (uint32_t)(rhi[e] - rlo[e]);
        const bool has = ok[e] && deg > 0 && (int64_t)deg < I;
        ei[e] = (uint32_t)I;
        if (has) ei[e] = (uint32_t)indices[rlo[e] + (int64_t)(((uint64_t)xorshift32(s) * deg) >> 32)];
    }
#ifdef RSX_ABLATE
    if (RSX_ABL(32) && g_mock_neg != nullptr) {
        for (int e0 = 0; e0 < kPerThread; e0 += 4) {             // four positions at a time: 96 registers of rows in flight
            int32_t row[4][24];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int deg = (int)(rhi[e0 + g] - rlo[e0 + g]);
#pragma unroll
                for (int q = 0; q < 24; ++q) row[g][q] = (ok[e0 + g] && q < deg) ? indices[rlo[e0 + g] + q] : -1;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t s2 = rng_seed(seed, step, (uint64_t)(piece_lo + (int64_t)blk * kChunk + (e0 + g) * kChunkThreads + tid), 0x5bd1e995ull);
                const uint32_t blk2 = mulhi32((uint32_t)splitmix64(eu[e0 + g] ^ ei[e0 + g]), (uint32_t)(I / 2));
                int32_t cand = 0;
                for (int tries = 0; tries < 8; ++tries) {
                    cand = (int32_t)(2u * (uint32_t)neg_block_of(blk2, I / 2, seed | 1ull) + (xorshift32(s2) & 1u));
                    bool in = false;
#pragma unroll
                    for (int q = 0; q < 24; ++q) in |= row[g][q] == cand;
                    if (!in) break;
                }
                const int64_t loc = (int64_t)blk * kChunk + (e0 + g) * kChunkThreads + tid;
                if (ok[e0 + g]) g_mock_neg[loc] = (uint32_t)cand;
            }
        }
    }
#endif
    uint32_t c0[kPerThread], c1[kPerThread];
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
        const uint32_t it = (int64_t)ei[e] < I ? ei[e] : 0u;
        const U32Pair cc = *reinterpret_cast<const U32Pair *>(cdf + it);
        c0[e] = cc.a; c1[e] = cc.b;
    }
#pragma unroll
    for (int e = 0; e < kPerThread; ++e) {
        eb[e] = 0xFFFFFFFFu;
        if (ok[e]) {
            int bk = nbm;                                              // "no positive": own last bucket
            if ((int64_t)ei[e] < I) {
                // a point of the item's CDF interval picked by a hash of the user (a popular item spreads over buckets)
                const uint32_t t = c0[e] + mulhi32((uint32_t)splitmix64(0xC2B2AE3D27D4EB4Full ^ eu[e]), c1[e] - c0[e]);
                if (ca.C > 1) {      // buckets of equal mass INSIDE the item's range: none straddles two ranges
                    // (monotone in t, which is all the buckets need: floor((t - lo) * NBc / mass) up to the rounding of the product)
                    const int ch = (int)(ei[e] / (uint32_t)ca.g.Ic);
                    uint32_t q = (uint32_t)((double)(t - rng_lo[ch]) * rng_scale[ch]);
                    if (q >= (uint32_t)ca.NBc) q = (uint32_t)ca.NBc - 1u;
                    bk = ch * ca.NBc + (int)q;
                } else {
                    bk = (int)mulhi32(t, (uint32_t)nbm);
                }
            }
            eb[e] = ((uint32_t)bk << 13) | atomicAdd(&hist[bk], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the NB counts: a contiguous slice per thread, then a block scan of the slice sums
    const int per = (NB + kChunkThreads - 1) / kChunkThreads;
    const int q0 = tid * per < NB ? tid * per : NB, q1 = (q0 + per < NB) ? q0 + per : NB;
    uint32_t mine = 0;
    for (int q = q0; q < q1; ++q) mine += hist[q];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_sum[w];
    for (int q = q0; q < q1; ++q) {
        const uint32_t cnt = hist[q];
        hist[q] = base;
        table[(size_t)blk * NB + q] = (base << 16) | cnt;     // cnt <= 4096, base < 4096 (row of this chunk: contiguous)
        if (cnt) atomicAdd(&totals[(size_t)q * kTotalStride], cnt);   // integer: order-independent; one counter per 128-B line
        base += cnt;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kPerThread; ++e)
        if (eb[e] != 0xFFFFFFFFu) stage[hist[eb[e] >> 13] + (eb[e] & 0x1FFFu)] = make_uint2(eu[e], ei[e]);
    __syncthreads();
    const int64_t chunk_n = (n - (int64_t)blk * kChunk < kChunk) ? n - (int64_t)blk * kChunk : kChunk;
    for (int q = tid; q < chunk_n; q += kChunkThreads) pairs[(size_t)blk * kChunk + q] = stage[q];
}

// ascending-only bitonic network (each merge starts with a mirror stage), so positions >= n act
// as +infinity without being stored: a pair whose upper index is >= n is skipped
template <class Acc>
__device__ __forceinline__ void bitonic_sort(Acc a, int n, int tid)
{
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += kBlock) {
                int lo, hi;
                if (j == (k >> 1)) { const int w = t & (j - 1); lo = ((t - w) << 1) + w; hi = lo + ((j - w) << 1) - 1; }
                else               { const int w = t & (j - 1); lo = ((t - w) << 1) + w; hi = lo + j; }
                if (hi < n) {
                    const uint64_t x = a.get(lo), y = a.get(hi);
                    if (x > y) { a.set(lo, y); a.set(hi, x); }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void sort2(uint64_t &x, uint64_t &y)
{
    const bool sw = y < x;
    const uint64_t lo = sw ? y : x, hi = sw ? x : y;
    x = lo; y = hi;
}

// the same network on LDS keys, two stages per pass: a thread owns the four keys that the stages
// j and j/2 (or the mirror stage and j = k/4) permute among themselves, so the LDS traffic, the
// index arithmetic and the barriers are halved (30 passes instead of 55 stages for 1024 keys).
// keys[n .. n2) are padded with +infinity (no (item, user) key is all ones).
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, int n, int tid)
{
    int n2 = 4;
    while (n2 < n) n2 <<= 1;
    for (int q = n + tid; q < n2; q += kBlock) keys[q] = ~0ull;
    __syncthreads();
    const int groups = n2 >> 2;
    for (int g = tid; g < groups; g += kBlock) {          // k = 2 and k = 4 inside four consecutive keys
        uint64_t *c = keys + 4 * g;
        uint64_t v0 = c[0], v1 = c[1], v2 = c[2], v3 = c[3];
        sort2(v0, v1); sort2(v2, v3);
        sort2(v0, v3); sort2(v1, v2);
        sort2(v0, v1); sort2(v2, v3);
        c[0] = v0; c[1] = v1; c[2] = v2; c[3] = v3;
    }
    __syncthreads();
    for (int k = 8; k <= n2; k <<= 1) {
        const int q = k >> 2;                             // mirror stage of this merge + stage j = k/4
        for (int g = tid; g < groups; g += kBlock) {
            const int low = g & (q - 1), base = (g - low) << 2;
            const int e0 = base + low, e1 = e0 + q, e3 = base + k - 1 - low, e2 = e3 - q;
            uint64_t v0 = keys[e0], v1 = keys[e1], v2 = keys[e2], v3 = keys[e3];
            sort2(v0, v3); sort2(v1, v2);
            sort2(v0, v1); sort2(v2, v3);
            keys[e0] = v0; keys[e1] = v1; keys[e2] = v2; keys[e3] = v3;
        }
        __syncthreads();
        int j = k >> 3;
        for (; j >= 4; j >>= 2) {                         // stages j and j/2
            const int h = j >> 1;
            for (int g = tid; g < groups; g += kBlock) {
                const int low = g & (h - 1), a = ((g - low) << 2) | low;
                uint64_t v0 = keys[a], v1 = keys[a + h], v2 = keys[a + 2 * h], v3 = keys[a + 3 * h];
                sort2(v0, v2); sort2(v1, v3);
                sort2(v0, v1); sort2(v2, v3);
                keys[a] = v0; keys[a + h] = v1; keys[a + 2 * h] = v2; keys[a + 3 * h] = v3;
            }
            __syncthreads();
        }
        if (j >= 1) {                                     // stages 2 and 1, or 1 alone: four consecutive keys
            for (int g = tid; g < groups; g += kBlock) {
                uint64_t *c = keys + 4 * g;
                uint64_t v0 = c[0], v1 = c[1], v2 = c[2], v3 = c[3];
                if (j == 2) { sort2(v0, v2); sort2(v1, v3); }
                sort2(v0, v1); sort2(v2, v3);
                c[0] = v0; c[1] = v1; c[2] = v2; c[3] = v3;
            }
            __syncthreads();
        }
    }
}

// a bucket too large for LDS is sorted in place in its own slice of u_out / i_out (one workgroup,
// device-scope relaxed accesses so nothing is cached in registers across the barriers)
struct GlobalKeys {
    int32_t *u, *i;
    __device__ __forceinline__ uint64_t get(int q) const
    {
        const uint32_t uu = (uint32_t)__hip_atomic_load(u + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t ii = (uint32_t)__hip_atomic_load(i + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((uint64_t)ii << 32) | uu;
    }
    __device__ __forceinline__ void set(int q, uint64_t v) const
    {
        __hip_atomic_store(u + q, (int32_t)(uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(i + q, (int32_t)(uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// negatives of kNegGroup batch positions per thread, advanced in lockstep (one dependent load
// level at a time: signature, row bounds, each binary-search probe), same draws as draw_negative()
constexpr int kNegGroup = 4;
static_assert(kMaxChunks % kBlock == 0, "bucket_sort_kernel reads kMaxChunks / kBlock chunk-table entries per thread");

// chunked layout (cx.C > 1): the ranges are those of the position's item range (pc = its first position, nc = its
// live positions), the fall-back candidates are the REAL items of that range -- never the whole catalog: a negative
// outside the range would reach G after the range was handed on -- and a user who owns the whole range gets no
// negative (nj = -1) after 192 tries.
struct NegCtx {
    int C, ch;
    ChunkGeom g;
    int64_t pc, nc;
    const uint8_t *wn;       // real items per negative range of the bucket's table (chunked layout)
    bool whole;              // chunked layout without blocks: every draw is over the range's real items
};

__device__ __forceinline__ void negatives_lockstep(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, const uint64_t *__restrict__ user_sig,
    int64_t I, int64_t B, uint64_t seed, uint64_t step, int neg_block, uint64_t neg_key, const int64_t *wstart,
    const int32_t *wlo, int m, const bool (&live)[kNegGroup], const int64_t (&p)[kNegGroup],
    const uint32_t (&u)[kNegGroup], int32_t (&nj)[kNegGroup], const NegCtx &cx)
{
    const bool chunked = cx.C > 1;
    const int64_t flo = chunked ? (int64_t)cx.ch * cx.g.Ic : 0, fn = chunked ? cx.g.real(cx.ch) : I;   // fall-back candidates
    const int give_up = chunked ? 192 : 0x7fffffff;
    int64_t nlo[kNegGroup], nn[kNegGroup], rlo[kNegGroup], rhi[kNegGroup];
    uint32_t s[kNegGroup];
    ulonglong2 rec[kNegGroup];
    bool need[kNegGroup];
    const bool use_sig = user_sig != nullptr && neg_block > 0;
#pragma unroll
    for (int g = 0; g < kNegGroup; ++g) {
        rec[g] = make_ulonglong2(~0ull, 0ull);
        if (live[g] && use_sig) rec[g] = RSX_ABL(8) ? make_ulonglong2(0ull, 0ull) : reinterpret_cast<const ulonglong2 *>(user_sig)[u[g]];   // (8: every block "proven" negative)
    }
    bool any = false;
#pragma unroll
    for (int g = 0; g < kNegGroup; ++g) {
        nj[g] = -1; need[g] = false; nlo[g] = 0; nn[g] = I; s[g] = 1u; rlo[g] = 0; rhi[g] = 0;
        if (live[g]) {
            s[g] = rng_seed(seed, step, (uint64_t)p[g], 0x5bd1e995ull);
            if (m > 0) {           // the bucket's table of (first position, item block) per negative range
                int lo = 0, hi = m;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wstart[mid] <= p[g]) lo = mid; else hi = mid; }
                nlo[g] = wlo[lo];
                nn[g] = chunked ? (int64_t)cx.wn[lo] : ((nlo[g] + neg_block <= I) ? neg_block : I - nlo[g]);
            } else if (chunked && cx.whole) {
                nlo[g] = flo; nn[g] = fn;
            } else if (chunked) {
                neg_range_chunk(cx.g, cx.ch, p[g] - cx.pc, cx.nc, neg_key, nlo[g], nn[g]);
            } else {
                neg_range(I, p[g], B, neg_block, neg_key, nlo[g], nn[g]);
            }
            const bool empty = nn[g] == 0;                 // (chunked layout: a block that is all padding)
            if (empty) { nlo[g] = flo; nn[g] = fn; }
            // a clear signature bit proves the whole item block negative for this user
            // (73 % of the draws at 20 positives per user): no row read at all
            need[g] = !use_sig || empty || ((rec[g].x >> sig_bit(nlo[g] / neg_block)) & 1ull) != 0ull;
            if (!need[g]) nj[g] = (int32_t)(nlo[g] + (int64_t)(((uint64_t)xorshift32(s[g]) * (uint64_t)nn[g]) >> 32));
            any |= need[g];
        }
    }
    if (!any) return;
    // the exact test.  The signature record also carries the row's start and length, so a short
    // row is fetched whole with independent loads (one latency instead of indptr + a chain of
    // binary-search probes) and every redraw is then tested in registers.
    constexpr int kRowRegs = 24;
    any = false;
#pragma unroll
    for (int g = 0; g < kNegGroup; ++g) {
        if (!need[g]) continue;
        const uint64_t len = rec[g].y >> 40;
        if (use_sig && len < kSigLenClip) { rlo[g] = (int64_t)(rec[g].y & kSigStartMask); rhi[g] = rlo[g] + (int64_t)len; }
        else { rlo[g] = indptr[u[g]]; rhi[g] = indptr[u[g] + 1]; }
        if (rhi[g] - rlo[g] <= kRowRegs) {
            const int32_t *row = indices + rlo[g];
            const int deg = (int)(rhi[g] - rlo[g]);
            int32_t v[kRowRegs];
#pragma unroll
            for (int q = 0; q < kRowRegs; ++q) v[q] = q < deg ? row[q] : -1;
            for (int tries = 0;; ++tries) {
                if (tries == 64) { nlo[g] = flo; nn[g] = fn; }  // the user owns (nearly) the whole block
                if (tries == give_up) break;                    // ... and the whole range: no negative (nj stays -1)
                const int32_t cand = (int32_t)(nlo[g] + (int64_t)(((uint64_t)xorshift32(s[g]) * (uint64_t)nn[g]) >> 32));
                bool in = false;
#pragma unroll
                for (int q = 0; q < kRowRegs; ++q) in |= v[q] == cand;
                if (!in) { nj[g] = cand; break; }
            }
            need[g] = false;
        }
        any |= need[g];
    }
    for (int tries = 0; any; ++tries) {                          // long rows: binary search, in lockstep
        int64_t a[kNegGroup], z[kNegGroup];
        int32_t cand[kNegGroup];
#pragma unroll
        for (int g = 0; g < kNegGroup; ++g) {
            a[g] = 0; z[g] = 0; cand[g] = 0;
            if (need[g] && tries == give_up) need[g] = false;   // owns the whole range: no negative (nj stays -1)
            if (need[g]) {
                if (tries == 64) { nlo[g] = flo; nn[g] = fn; }
                cand[g] = (int32_t)(nlo[g] + (int64_t)(((uint64_t)xorshift32(s[g]) * (uint64_t)nn[g]) >> 32));
                a[g] = rlo[g]; z[g] = rhi[g];
            }
        }
        for (;;) {                                               // lower_bound of cand in the sorted row
            bool more = false;
            int32_t v[kNegGroup];
            int64_t mid[kNegGroup];
#pragma unroll
            for (int g = 0; g < kNegGroup; ++g) {
                mid[g] = (a[g] + z[g]) >> 1; v[g] = 0;
                if (a[g] < z[g]) { v[g] = indices[mid[g]]; more = true; }
            }
            if (!more) break;
#pragma unroll
            for (int g = 0; g < kNegGroup; ++g)
                if (a[g] < z[g]) { if (v[g] < cand[g]) a[g] = mid[g] + 1; else z[g] = mid[g]; }
        }
        int32_t at[kNegGroup];
#pragma unroll
        for (int g = 0; g < kNegGroup; ++g) at[g] = (need[g] && a[g] < rhi[g]) ? indices[a[g]] : -1;
        any = false;
#pragma unroll
        for (int g = 0; g < kNegGroup; ++g)
            if (need[g]) {
                if (at[g] != cand[g]) { nj[g] = cand[g]; need[g] = false; }
                any |= need[g];
            }
    }
}

__global__ __launch_bounds__(kBlock, 5) void bucket_sort_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, const uint64_t *__restrict__ user_sig,
    int64_t I, int64_t B, int64_t piece_lo, uint64_t seed, uint64_t step, int neg_block, uint64_t neg_key, int nbm,
    int nblk, int sort_cap, const uint2 *__restrict__ pairs, const uint32_t *__restrict__ table,
    const uint32_t *__restrict__ totals, int32_t *u_out, int32_t *i_out, int32_t *__restrict__ j_out, ChunkArgs ca)
{
    __shared__ uint64_t keys[kSortCap];
    __shared__ uint32_t cstart[kMaxChunks + 1];    // first bucket-local rank of each chunk's slice
    __shared__ uint16_t csrc[kMaxChunks];          // where that slice starts inside the chunk
    __shared__ uint32_t red[kBlock / 64];
    __shared__ uint32_t wsum[kBlock / 64];
    __shared__ uint32_t red_cb[kBlock / 64], red_ct[kBlock / 64];
    __shared__ int64_t wstart[kRangeCap];          // first batch position of each negative range met here
    __shared__ int32_t wlo[kRangeCap];             // and the item block it draws from
    __shared__ uint8_t wn[kRangeCap];              // and (chunked layout) the REAL items in that block
    const int tid = threadIdx.x;
    const int bk = blockIdx.x;
    const int n = (int)totals[(size_t)bk * kTotalStride];
    const bool chunked = ca.C > 1;
    if (n == 0 && !chunked) return;
    // first output position of this bucket = pairs in the buckets before it; chunked layout: also the first position
    // (cb) and the number of live positions (ct) of the item range this bucket belongs to
    const int ch = chunked ? (bk < nbm ? bk / ca.NBc : ca.C - 1) : 0;
    uint32_t before = 0, cb = 0, ct = 0;
    if (chunked) {
        const int r0 = ch * ca.NBc, r1 = (bk < nbm) ? r0 + ca.NBc : nbm;
        for (int q = tid; q < r1; q += kBlock) {
            const uint32_t v = totals[(size_t)q * kTotalStride];
            if (q < bk) before += v;
            if (q < r0) cb += v; else ct += v;
        }
    } else
    for (int q = tid; q < bk; q += kBlock) before += totals[(size_t)q * kTotalStride];
    // this bucket's slice of every chunk: (offset in chunk, count) -> exclusive scan of the counts
    constexpr int kEnt = kMaxChunks / kBlock;      // 2 table entries per thread
    uint32_t ent[kEnt], mine = 0;
#pragma unroll
    for (int e = 0; e < kEnt; ++e) {
        const int blk = tid * kEnt + e;
        ent[e] = blk < nblk ? table[(size_t)blk * (nbm + 1) + bk] : 0u;
        mine += ent[e] & 0xFFFFu;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off); cb += __shfl_xor(cb, off); ct += __shfl_xor(ct, off); }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 0) { red[tid >> 6] = before; red_cb[tid >> 6] = cb; red_ct[tid >> 6] = ct; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
#pragma unroll
    for (int e = 0; e < kEnt; ++e) {
        const int blk = tid * kEnt + e;
        cstart[blk] = base; csrc[blk] = (uint16_t)(ent[e] >> 16);
        base += ent[e] & 0xFFFFu;
    }
    if (tid == kBlock - 1) cstart[kMaxChunks] = base;
    const int64_t p0 = piece_lo + red[0] + red[1] + red[2] + red[3];
    NegCtx cx{ca.C, ch, ca.g, 0, 0, wn, ca.whole};
    if (chunked) {
        cx.pc = (int64_t)red_cb[0] + red_cb[1] + red_cb[2] + red_cb[3];
        cx.nc = (int64_t)red_ct[0] + red_ct[1] + red_ct[2] + red_ct[3];
        // the first bucket of a range publishes the range's first position; the "no positive" bucket the number of live ones
        if (tid == 0 && bk < nbm && bk % ca.NBc == 0) ca.chunk_pos_out[ch] = cx.pc;
        if (tid == 0 && bk == nbm) ca.chunk_pos_out[ca.C] = p0;
        if (n == 0) return;
    }
    const bool in_lds = n <= sort_cap;
    int32_t *ug = u_out + p0, *ig = i_out + p0;
    // negative ranges (rsx.h: neg_block) the positions [p0, p0 + n) fall in: range w starts at
    // position ceil(w*c*B/I) and draws from item block pi(w); 64-bit divisions and the block
    // permutation are paid once per range here instead of once per position
    int m = 0;
    if (chunked && ca.whole) {
        // (no blocks: the candidates of every position are the real items of its range)
    } else if (chunked && bk < nbm) {
        const ChunkGeom &g = ca.g;
        const int64_t w_first = (((p0 - cx.pc) * g.Ic) / cx.nc) / g.c, w_last = (((p0 + n - 1 - cx.pc) * g.Ic) / cx.nc) / g.c;
        if (w_last - w_first < kRangeCap) {
            m = (int)(w_last - w_first) + 1;
            const uint64_t key = chunk_key(neg_key, ch);
            for (int q = tid; q < m; q += kBlock) {
                const int64_t blk = neg_block_of(w_first + q, g.nbc, key);
                wstart[q] = cx.pc + ceil_div64((w_first + q) * g.c * cx.nc, g.Ic);
                wlo[q] = (int32_t)((int64_t)ch * g.Ic + blk * g.c);
                const int64_t left = g.real(ch) - blk * g.c;
                wn[q] = (uint8_t)(left <= 0 ? 0 : (left < g.c ? left : g.c));
            }
        }
    } else if (neg_block > 0) {
        const int64_t w_first = ((p0 * I) / B) / neg_block, w_last = (((p0 + n - 1) * I) / B) / neg_block;
        if (w_last - w_first < kRangeCap) {
            m = (int)(w_last - w_first) + 1;
            const int64_t nblocks = ceil_div64(I, neg_block);
            for (int q = tid; q < m; q += kBlock) {
                wstart[q] = ceil_div64((w_first + q) * neg_block * B, I);
                wlo[q] = (int32_t)(neg_block_of(w_first + q, nblocks, neg_key) * neg_block);
            }
        }
    }
    __syncthreads();
    // gather: rank r of the bucket lives in the chunk whose slice covers r
    // (kGather ranks per thread in lockstep: their pair loads are in flight together -- a bucket of ~832 pairs is one trip)
    constexpr int kGather = 4;
    for (int r0 = 0; r0 < n; r0 += kBlock * kGather) {
        uint32_t src[kGather];
#pragma unroll
        for (int e = 0; e < kGather; ++e) {
            const int r = r0 + e * kBlock + tid;
            int lo = 0, hi = kMaxChunks;                           // last chunk with cstart <= r
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cstart[mid] <= (uint32_t)r) lo = mid; else hi = mid; }
            src[e] = (uint32_t)lo * kChunk + csrc[lo] + ((uint32_t)r - cstart[lo]);
        }
        uint2 pr[kGather];
#pragma unroll
        for (int e = 0; e < kGather; ++e) pr[e] = (r0 + e * kBlock + tid < n && !RSX_ABL(16)) ? pairs[src[e]] : make_uint2(0u, 0u);
#pragma unroll
        for (int e = 0; e < kGather; ++e) {
            const int r = r0 + e * kBlock + tid;
#ifdef RSX_ABLATE
            if (RSX_ABL(32) && g_mock_neg != nullptr && r < n) { const uint32_t w4 = g_mock_neg[src[e]]; asm volatile("" :: "v"(w4)); }      // (the mock's fourth word: gathered, not used)
#endif
            const uint64_t kv = ((uint64_t)pr[e].y << 32) | pr[e].x;
            if (r < n) { if (in_lds) keys[r] = kv; else GlobalKeys{ug, ig}.set(r, kv); }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (in_lds) {
        if (!RSX_ABL(4)) bitonic_sort_lds(keys, n, tid);
    } else if (bk != nbm) {        // (the "no positive" bucket holds one key; its order is irrelevant)
        bitonic_sort(GlobalKeys{ug, ig}, n, tid);
    }
    for (int r0 = 0; r0 < n; r0 += kBlock * kNegGroup) {
        bool live[kNegGroup];
        int64_t p[kNegGroup];
        uint32_t u[kNegGroup], item[kNegGroup];
        int32_t nj[kNegGroup];
#pragma unroll
        for (int g = 0; g < kNegGroup; ++g) {
            const int r = r0 + g * kBlock + tid;
            uint64_t kv = ~0ull;
            if (r < n) kv = in_lds ? keys[r] : GlobalKeys{ug, ig}.get(r);
            u[g] = (uint32_t)kv; item[g] = (uint32_t)(kv >> 32);
            p[g] = p0 + r;
            live[g] = r < n && (int64_t)item[g] < I;
        }
        negatives_lockstep(indptr, indices, user_sig, I, B, seed, step, neg_block, neg_key, wstart, wlo, m, live, p, u, nj, cx);
#pragma unroll
        for (int g = 0; g < kNegGroup; ++g) {
            const int r = r0 + g * kBlock + tid;
            // (a live pair without a negative -- chunked layout, the user owns its whole range -- is skipped by the step)
            if (r < n) { u_out[p[g]] = (int32_t)u[g]; i_out[p[g]] = (live[g] && nj[g] >= 0) ? (int32_t)item[g] : -1; j_out[p[g]] = nj[g]; }
        }
    }
}

// ---- whole-pass batches without bucketing or sorting: the CSC walk -------------------------------
// A batch that holds EVERY user once (batch == num_users: the reference's epoch, data/generators.py:206-210, and the shape of
// every BASELINE config's bench step) needs no permutation, no buckets and no sort to come out ordered by positive item: the
// transposed interaction matrix (CSC: item -> its users, ascending) IS that order.  Per entry e = (user u, rank of the item inside
// u's row, deg(u)) the blob built by rsx_bpr_build_csc holds 6 bytes (8 when a row is longer than 255); one streaming pass keeps
// entry e iff  pick(seed, step, u) == rank  -- exactly one positive per user, uniform in its row -- and compacts the kept
// (user, item) pairs IN ORDER (ballots inside a wavefront, a 32-entry LDS scan inside the workgroup, decoupled look-back between
// the tiles), so the output is ordered by (item, user) with no sort at all.  A second kernel draws the negatives per ordered
// position exactly like the bucket path (negatives_lockstep: signature first, rejection against the row).
// Against the bucket path (two latency-bound passes, 0.34 GB, 105 us alone / 300 us beside the step kernel at the headline shape)
// this is one streaming read of 6 bytes per interaction (120 MB) plus the negatives.
constexpr int kCscThreads = 256;
// rounds of one uint4 of users per thread: 8 -> 8192 entries per workgroup, 110 VGPRs; 4 -> 4096 entries, fewer registers -- the
// kernel runs in what the step kernel's wavefronts leave free of a SIMD's registers (development A/B: -DRSX_CSC_ROUNDS=2 / 4 / 8)
#ifndef RSX_CSC_ROUNDS
#define RSX_CSC_ROUNDS 4
#endif
constexpr int kCscRounds = RSX_CSC_ROUNDS;
constexpr int kCscTile = kCscThreads * 4 * kCscRounds;         // entries per workgroup
constexpr int kCscCounts = kCscRounds * (kCscThreads / 64);    // (round, wavefront) counts of a tile
constexpr int kCscItemsLds = 2048;                             // item borders of a tile kept in LDS (more: searched in memory)
constexpr int kCscNegBlock = kBlock * kNegGroup;               // positions per workgroup of the negatives pass
static_assert(kCscThreads == 256 && kCscCounts <= 32 && kCscTile <= 65535, "the tile's (round, wavefront) counts are scanned by half a wavefront; borders are uint16");

// the positive of user u in this step: rank in [0, deg).  Keyed per (seed, step); two multiplies -- 20M entries per step run it
__host__ __device__ __forceinline__ uint32_t csc_hash(uint32_t u, uint32_t k0, uint32_t k1)
{
    uint32_t x = u ^ k0;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x ^ k1;
}
__host__ __device__ __forceinline__ uint64_t csc_step_key(uint64_t seed, uint64_t step) { return splitmix64(seed ^ (step * 0x9E3779B97F4A7C15ull) ^ 0xC5C0DE5A3D1E7ull); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename RD> struct RdBits;
template <> struct RdBits<uint16_t> { static constexpr int shift = 8; static constexpr uint32_t mask = 0xFFu; };
template <> struct RdBits<uint32_t> { static constexpr int shift = 16; static constexpr uint32_t mask = 0xFFFFu; };


struct CscArgs {
    const int64_t *ptr;          // [I + 1] first entry of every item
    const int32_t *tile_item;    // [ntiles + 1] the item that holds the first entry of every tile (last: I - 1)
    const uint32_t *users;       // [ntiles * kCscTile]
    const void *rd;              // rank | deg << shift per entry
    int64_t nnz, I;
    int ntiles;
};

// pass 1 of the walk: one workgroup per tile of kCscTile entries.  No workgroup waits for another: the kept pairs of a tile go, in
// order, to the tile's OWN slice of a staging buffer and its count to counts[tile]; a one-workgroup scan turns the counts into the
// tiles' first positions and a copy pass moves the slices into place.  (The first form of this kernel compacted in ONE pass with a
// decoupled look-back between the tiles: 85 us alone for the 120 MB of the headline shape -- a tile's look-back is a chain of
// device-scope loads across the XCDs' L2s, ~1.5 us per 64 tiles, and the ~1 800 resident tiles of the first round all wait for it
// while holding their wave slots.  profiles/r06_exp_csc_sampler.txt)
template <typename RD>
__global__ __launch_bounds__(kCscThreads) void csc_select_kernel(CscArgs a, uint32_t k0, uint32_t k1, uint2 *__restrict__ stage,
                                                                 uint32_t *__restrict__ counts, int C, int64_t Ic,
                                                                 uint32_t *__restrict__ border_tile, uint32_t *__restrict__ border_rank)
{
    __shared__ uint32_t s_cnt[kCscCounts], s_ex[kCscCounts + 1];
    __shared__ uint16_t s_border[kCscItemsLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int64_t e0 = (int64_t)tile * kCscTile;
    // (non-temporal: 120 MB stream through once per step and must not push the item table out of the memory-side cache)
    const u32x4 *up = reinterpret_cast<const u32x4 *>(a.users) + e0 / 4;
    u32x4 us[kCscRounds];
    uint32_t rdv[kCscRounds][4];
#pragma unroll
    for (int k = 0; k < kCscRounds; ++k) us[k] = __builtin_nontemporal_load(up + k * kCscThreads + tid);
    if constexpr (sizeof(RD) == 2) {
        const u32x2 *rp = reinterpret_cast<const u32x2 *>(a.rd) + e0 / 4;
#pragma unroll
        for (int k = 0; k < kCscRounds; ++k) {
            const u32x2 v = __builtin_nontemporal_load(rp + k * kCscThreads + tid);
            rdv[k][0] = v.x & 0xFFFFu; rdv[k][1] = v.x >> 16; rdv[k][2] = v.y & 0xFFFFu; rdv[k][3] = v.y >> 16;
        }
    } else {
        const u32x4 *rp = reinterpret_cast<const u32x4 *>(a.rd) + e0 / 4;
#pragma unroll
        for (int k = 0; k < kCscRounds; ++k) {
            const u32x4 v = __builtin_nontemporal_load(rp + k * kCscThreads + tid);
            rdv[k][0] = v.x; rdv[k][1] = v.y; rdv[k][2] = v.z; rdv[k][3] = v.w;
        }
    }
    // the item borders inside this tile (while the loads above travel): item(x) = first + #{borders <= x}
    const int first = a.tile_item[tile], nb = a.tile_item[tile + 1] - first;
    const bool borders_in_lds = nb <= kCscItemsLds;
    if (borders_in_lds)
        for (int q = tid; q < nb; q += kCscThreads) {
            const int64_t b = a.ptr[first + 1 + q] - e0;
            s_border[q] = (uint16_t)(b < kCscTile ? b : kCscTile);      // (a border at or beyond the tile's end is never <= x)
        }
    // keep flags, in entry order: round k, thread tid, element j  <->  entry e0 + (k * 256 + tid) * 4 + j
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    uint32_t keep = 0u;                                            // bit k * 4 + j
    uint32_t before[kCscRounds];                                   // kept entries of this round in the lanes below me (this wavefront)
#pragma unroll
    for (int k = 0; k < kCscRounds; ++k) {
        const uint32_t uu[4] = {us[k].x, us[k].y, us[k].z, us[k].w};
        const int64_t e = e0 + (int64_t)(k * kCscThreads + tid) * 4;
        uint32_t bef = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t rank = rdv[k][j] & RdBits<RD>::mask, deg = rdv[k][j] >> RdBits<RD>::shift;
            const bool kp = e + j < a.nnz && (int64_t)deg < a.I && mulhi32(csc_hash(uu[j], k0, k1), deg) == rank;
            const uint64_t m = __ballot(kp);
            bef += (uint32_t)__popcll(m & lanes_below);
            tot += (uint32_t)__popcll(m);
            keep |= (kp ? 1u : 0u) << (k * 4 + j);
        }
        before[k] = bef;
        if (lane == 0) s_cnt[k * 4 + wave] = tot;
    }
    __syncthreads();
    if (tid < kCscCounts) {                                        // exclusive scan of the (round, wavefront) counts
        const uint32_t mine = s_cnt[tid];
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < kCscCounts; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (tid >= off) incl += v;
        }
        s_ex[tid] = incl - mine;
        if (tid == kCscCounts - 1) { s_ex[kCscCounts] = incl; counts[tile] = incl; }
    }
    __syncthreads();
    // rank (inside the tile) of the entry (k, j) of this thread
    auto rank_of = [&](int k, int j) -> uint32_t {
        return s_ex[k * 4 + wave] + before[k] + (uint32_t)__popc(keep & (((1u << j) - 1u) << (k * 4)));
    };
    if (keep) {
        uint2 *mine = stage + e0;                                  // the tile's slice: room for every entry of the tile
#pragma unroll
        for (int k = 0; k < kCscRounds; ++k) {
            if (!((keep >> (k * 4)) & 0xFu)) continue;
            const uint32_t uu[4] = {us[k].x, us[k].y, us[k].z, us[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((keep >> (k * 4 + j)) & 1u)) continue;
                const int x = (k * kCscThreads + tid) * 4 + j;
                int item;
                if (borders_in_lds) {
                    int lo = 0, hi = nb;                           // #{q : border[q] <= x}
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)s_border[mid] <= x) lo = mid + 1; else hi = mid; }
                    item = first + lo;
                } else {
                    int lo = first, hi = first + nb;               // last item whose first entry is <= e0 + x
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.ptr[mid] <= e0 + x) lo = mid; else hi = mid - 1; }
                    item = lo;
                }
                mine[rank_of(k, j)] = make_uint2(uu[j], (uint32_t)item);
            }
        }
    }
    // where the item ranges start (chunked layout): the tile that holds the first entry of range ch says how many of its kept
    // entries lie in front of it; the scan adds the tile's first position
    for (int ch = 1; ch < C; ++ch) {
        const int64_t eb = a.ptr[ch * Ic];
        if (eb >= a.nnz) { if (tile == 0 && tid == 0) { border_tile[ch] = (uint32_t)a.ntiles; border_rank[ch] = 0u; } continue; }
        if (eb < e0 || eb >= e0 + kCscTile) continue;
        const int x = (int)(eb - e0), k = x / (kCscThreads * 4), t = (x / 4) % kCscThreads, j = x % 4;
        if (t == tid) {
            uint32_t r = 0;
#pragma unroll
            for (int kk = 0; kk < kCscRounds; ++kk) if (kk == k) r = rank_of(kk, j);
            border_tile[ch] = (uint32_t)tile; border_rank[ch] = r;
        }
    }
}

// pass 2: one workgroup scans the tiles' counts into their first positions; the number of live positions and the ranges' first positions
constexpr int kCscScanThreads = 1024;
__global__ __launch_bounds__(kCscScanThreads) void csc_scan_kernel(const uint32_t *__restrict__ counts, uint32_t *__restrict__ first_pos, int ntiles,
                                                                   int64_t *__restrict__ n_live_out)
{
    __shared__ uint32_t wsum[kCscScanThreads / 64];
    const int tid = threadIdx.x;
    const int per = (ntiles + kCscScanThreads - 1) / kCscScanThreads;
    const int q0 = tid * per < ntiles ? tid * per : ntiles, q1 = (q0 + per < ntiles) ? q0 + per : ntiles;
    uint32_t mine = 0;
    for (int q = q0; q < q1; ++q) mine += counts[q];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) run += wsum[w];
    for (int q = q0; q < q1; ++q) { first_pos[q] = run; run += counts[q]; }
    if (tid == kCscScanThreads - 1) { first_pos[ntiles] = run; *n_live_out = (int64_t)run; }
}

// pass 3: every tile moves its slice of the staging buffer into place (the ranges' first positions are finished here: they need the tiles' first positions)
__global__ __launch_bounds__(kCscThreads) void csc_place_kernel(const uint2 *__restrict__ stage, const uint32_t *__restrict__ counts,
                                                                const uint32_t *__restrict__ first_pos, int ntiles, int32_t *__restrict__ u_out,
                                                                int32_t *__restrict__ i_out, int C, const uint32_t *__restrict__ border_tile,
                                                                const uint32_t *__restrict__ border_rank, int64_t *__restrict__ chunk_pos_out)
{
    const int tile = blockIdx.x;
    const uint32_t n = counts[tile], p0 = first_pos[tile];
    const uint2 *mine = stage + (int64_t)tile * kCscTile;
    for (uint32_t r = threadIdx.x; r < n; r += kCscThreads) {
        const uint2 v = mine[r];
        u_out[p0 + r] = (int32_t)v.x; i_out[p0 + r] = (int32_t)v.y;
    }
    if (chunk_pos_out != nullptr && threadIdx.x == 0)
        for (int ch = 1; ch < C; ++ch) {
            const uint32_t bt = border_tile[ch];
            if (bt == (uint32_t)tile) chunk_pos_out[ch] = (int64_t)p0 + border_rank[ch];
            else if (bt >= (uint32_t)ntiles && tile == 0) chunk_pos_out[ch] = (int64_t)first_pos[ntiles];
        }
    if (chunk_pos_out != nullptr && tile == 0 && threadIdx.x == 0) { chunk_pos_out[0] = 0; chunk_pos_out[C] = (int64_t)first_pos[ntiles]; }
}

// negatives of the ordered positions (the second half of bucket_sort_kernel on pairs that are already in place): a workgroup
// takes kCscNegBlock consecutive positions of ONE item range, or of the dead tail [n_live, B)
__global__ __launch_bounds__(kBlock, 5) void csc_neg_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, const uint64_t *__restrict__ user_sig,
    int64_t I, int64_t B, uint64_t seed, uint64_t step, int neg_block, uint64_t neg_key, const int64_t *__restrict__ n_live_dev,
    int32_t *u_out, int32_t *i_out, int32_t *__restrict__ j_out, ChunkArgs ca)
{
    __shared__ int64_t wstart[kRangeCap];
    __shared__ int32_t wlo[kRangeCap];
    __shared__ uint8_t wn[kRangeCap];
    const int tid = threadIdx.x;
    const bool chunked = ca.C > 1;
    const int64_t n_live = *n_live_dev;
    // which range this workgroup works in: ranges 0 .. C - 1 (the whole live batch when not chunked), then the dead tail
    int64_t blocks_before = 0, lo_pos = 0, hi_pos = 0;
    int ch = -1;
    const int R = chunked ? ca.C : 1;
    for (int k = 0; k <= R; ++k) {
        const int64_t a = k < R ? (chunked ? ca.chunk_pos_out[k] : 0) : n_live;
        const int64_t z = k < R ? (chunked ? ca.chunk_pos_out[k + 1] : n_live) : B;
        const int64_t nblk = ceil_div64(z - a, kCscNegBlock);
        if ((int64_t)blockIdx.x < blocks_before + nblk) { ch = k; lo_pos = a; hi_pos = z; break; }
        blocks_before += nblk;
    }
    if (ch < 0) return;
    const int64_t p0 = lo_pos + ((int64_t)blockIdx.x - blocks_before) * kCscNegBlock;
    const int n = (int)((hi_pos - p0 < kCscNegBlock) ? hi_pos - p0 : kCscNegBlock);
    if (ch == R) {                                                 // users without a usable row: no triplet (the step skips i < 0)
        for (int r = tid; r < n; r += kBlock) { u_out[p0 + r] = 0; i_out[p0 + r] = -1; j_out[p0 + r] = -1; }
        return;
    }
    NegCtx cx{ca.C, ch, ca.g, lo_pos, hi_pos - lo_pos, wn, ca.whole};
    int m = 0;
    if (chunked && ca.whole) {
    } else if (chunked) {
        const ChunkGeom &g = ca.g;
        const int64_t w_first = (((p0 - cx.pc) * g.Ic) / cx.nc) / g.c, w_last = (((p0 + n - 1 - cx.pc) * g.Ic) / cx.nc) / g.c;
        if (w_last - w_first < kRangeCap) {
            m = (int)(w_last - w_first) + 1;
            const uint64_t key = chunk_key(neg_key, ch);
            for (int q = tid; q < m; q += kBlock) {
                const int64_t blk = neg_block_of(w_first + q, g.nbc, key);
                wstart[q] = cx.pc + ceil_div64((w_first + q) * g.c * cx.nc, g.Ic);
                wlo[q] = (int32_t)((int64_t)ch * g.Ic + blk * g.c);
                const int64_t left = g.real(ch) - blk * g.c;
                wn[q] = (uint8_t)(left <= 0 ? 0 : (left < g.c ? left : g.c));
            }
        }
    } else if (neg_block > 0) {
        const int64_t w_first = ((p0 * I) / B) / neg_block, w_last = (((p0 + n - 1) * I) / B) / neg_block;
        if (w_last - w_first < kRangeCap) {
            m = (int)(w_last - w_first) + 1;
            const int64_t nblocks = ceil_div64(I, neg_block);
            for (int q = tid; q < m; q += kBlock) {
                wstart[q] = ceil_div64((w_first + q) * neg_block * B, I);
                wlo[q] = (int32_t)(neg_block_of(w_first + q, nblocks, neg_key) * neg_block);
            }
        }
    }
    __syncthreads();
    bool live[kNegGroup];
    int64_t p[kNegGroup];
    uint32_t u[kNegGroup];
    int32_t nj[kNegGroup];
#pragma unroll
    for (int g = 0; g < kNegGroup; ++g) {
        const int r = g * kBlock + tid;
        live[g] = r < n;
        p[g] = p0 + r;
        u[g] = live[g] ? (uint32_t)u_out[p[g]] : 0u;
    }
    negatives_lockstep(indptr, indices, user_sig, I, B, seed, step, neg_block, neg_key, wstart, wlo, m, live, p, u, nj, cx);
#pragma unroll
    for (int g = 0; g < kNegGroup; ++g)
        if (live[g]) {
            j_out[p[g]] = nj[g];
            if (nj[g] < 0) i_out[p[g]] = -1;                       // (chunked layout: the user owns its whole range -- no triplet)
        }
}

// ---- building the CSC blob (once per CSR) -----------------------------------------------------
__global__ __launch_bounds__(kBlock) void csc_entries_kernel(const int64_t *__restrict__ indptr, int64_t U, int shift,
                                                             uint64_t *__restrict__ vals, unsigned long long *__restrict__ max_deg)
{
    unsigned long long mx = 0;
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < U; u += (int64_t)gridDim.x * kBlock) {
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        const uint64_t deg = (uint64_t)(hi - lo);
        if (deg > mx) mx = deg;
        if (vals != nullptr)
            for (int64_t q = lo; q < hi; ++q) vals[q] = (uint64_t)(uint32_t)u | ((((uint64_t)(q - lo)) | (deg << shift)) << 32);
    }
    if (vals == nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
        if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(max_deg, mx);
    }
}

template <typename RD>
__global__ __launch_bounds__(kBlock) void csc_split_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, int64_t nnz,
                                                           int64_t nnz_pad, int64_t I, uint32_t *__restrict__ users, RD *__restrict__ rd,
                                                           int64_t *__restrict__ ptr, int32_t *__restrict__ tile_item, int ntiles)
{
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < nnz_pad; s += (int64_t)gridDim.x * kBlock) {
        if (s >= nnz) { users[s] = 0u; rd[s] = (RD)(RdBits<RD>::mask | (1u << RdBits<RD>::shift)); continue; }   // padding: rank >= deg, never kept
        const uint64_t v = vals[s];
        users[s] = (uint32_t)v;
        rd[s] = (RD)(v >> 32);
        const int64_t k = (int64_t)keys[s];
        if (s % kCscTile == 0) tile_item[s / kCscTile] = (int32_t)k;
        const int64_t kprev = s == 0 ? -1 : (int64_t)keys[s - 1];
        for (int64_t it = kprev + 1; it <= k; ++it) ptr[it] = s;    // first entry of item k and of the empty items in front of it
        if (s == nnz - 1) for (int64_t it = k + 1; it <= I; ++it) ptr[it] = nnz;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        tile_item[ntiles] = (int32_t)(I - 1);
        if (nnz == 0) { for (int64_t it = 0; it <= I; ++it) ptr[it] = 0; tile_item[0] = 0; }
    }
}

// ---- item CDF of the positive-sampling distribution (static per CSR) -------------------------
// mass[i] = sum over users holding i of floor(2^32 / deg(u)) (64-bit integer atomics: exact and
// order-independent), cdf[i] = floor(2^32 * prefix(mass)[i] / total), cdf[I] = 2^32 - 1
__global__ __launch_bounds__(kBlock) void item_mass_kernel(const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int64_t U, int64_t I,
                                                           unsigned long long *__restrict__ mass)
{
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < U; u += (int64_t)gridDim.x * kBlock) {
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        if (hi <= lo || hi - lo >= I) continue;                // never sampled (see bucket_chunk_kernel)
        const unsigned long long w = 0x100000000ull / (unsigned long long)(hi - lo);
        for (int64_t q = lo; q < hi; ++q) atomicAdd(&mass[indices[q]], w);
    }
}

constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void item_cdf_kernel(const unsigned long long *__restrict__ mass, int64_t I,
                                                              uint32_t *__restrict__ cdf)
{
    __shared__ unsigned long long part[kScanBlock];
    const int tid = threadIdx.x;
    const int64_t per = (I + kScanBlock - 1) / kScanBlock;
    const int64_t q0 = tid * per, q1 = (q0 + per < I) ? q0 + per : I;
    unsigned long long mine = 0;
    for (int64_t q = q0; q < q1; ++q) mine += mass[q];
    part[tid] = mine;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {          // Hillis-Steele inclusive scan
        const unsigned long long v = (tid >= off) ? part[tid - off] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const double total = (double)part[kScanBlock - 1];
    unsigned long long run = part[tid] - mine;
    for (int64_t q = q0; q < q1; ++q) {
        double f = total > 0.0 ? floor((double)run * 4294967296.0 / total) : 0.0;
        cdf[q] = f >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)f;
        run += mass[q];
    }
    if (tid == 0) cdf[I] = 0xFFFFFFFFu;
}

#ifdef RSX_ABLATE
// (mask 32, a MOCK of the redesign sketched in DESIGN.md section 9: the bucketing pass also loads the user's row -- four positions at a time,
//  24 registers each -- tests a candidate negative against it and writes a fourth word per pair; the sort pass gathers that word.  With 8:
//  what a sampler that draws the negative where the row is would cost the step.  Wrong triplets: timing only.)
// dev build only (librsx_dev.so), TIMING ONLY -- the triplets these switches leave behind are wrong: which part of the sampler disturbs the
// step kernel it runs beside (tools/sampler_parts.sh; the loop steps on replayed batches meanwhile: rsx_debug_set_sampler_replay(2))
//   1  no bucketing pass (the sort pass works on the chunks an earlier step left in the workspace)     2  no sort pass
//   4  the sort pass without its LDS sort      8  ... without the rejection test (no signature, no row)   16  ... without the gather of its pairs
int g_sample_ablate = 0;          // in force (host side)
int g_sample_ablate_wanted = 0;
#define RSX_SAMPLE_ABL_HOST(mask) ((g_sample_ablate & (mask)) != 0)
#else
#define RSX_SAMPLE_ABL_HOST(mask) false
#endif

unsigned grid_1d(int64_t n)
{
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rsx_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

int key_bits(int64_t num_items)   // keys are in [0, num_items]
{
    int bits = 1;
    while ((1ll << bits) <= num_items) ++bits;
    return bits;
}

size_t sort_temp_bytes(int64_t batch, int bits)
{
    size_t n = 0;
    uint32_t *nul = nullptr;
    (void)rocprim::radix_sort_pairs<sort_config>(nullptr, n, nul, nul, nul, nul, (size_t)batch, 0, (unsigned)bits, (hipStream_t)0);
    return n;
}

int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

}  // namespace

#ifdef RSX_ABLATE
RSX_API int rsx_debug_set_sample_ablation(int mask) { g_sample_ablate_wanted = mask; return RSX_OK; }
// armed by the native loop when its FIRST shadow sample is queued (the batches the loop replays were sampled whole); the caller has
// drained the sampler's stream: no sampler kernel is in flight when the device-side switch flips
int rsx_debug_sample_ablation_arm(bool on)
{
    const int mask = on ? g_sample_ablate_wanted : 0;
    if (mask == g_sample_ablate) return RSX_OK;
    g_sample_ablate = mask;
    if (mask & 32) {
        static uint32_t *scratch = nullptr;
        if (scratch == nullptr && hipMalloc((void **)&scratch, (size_t)kPiece * 4) != hipSuccess) return RSX_E_HIP;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_mock_neg), &scratch, sizeof(scratch)) != hipSuccess) return RSX_E_HIP;
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_rsx_ablate), &mask, sizeof(int)) == hipSuccess ? RSX_OK : RSX_E_HIP;
}
#endif

RSX_API int rsx_bpr_build_signature(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                                    int neg_block, uint64_t *sig_out, rsx_stream_t stream)
{
    RSX_CHECK_ARG(indptr_dev && indices_dev && sig_out, "null pointer");
    RSX_CHECK_ARG(num_users > 0 && neg_block >= 1 && neg_block <= kMaxNegBlock, "bad shape");
    hipLaunchKernelGGL(build_signature_kernel, dim3(grid_1d(num_users)), dim3(kBlock), 0, (hipStream_t)stream,
                       indptr_dev, indices_dev, num_users, neg_block, sig_out);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

// bucketed path: one piece of at most kPiece positions at a time
struct BucketWs { uint2 *pairs; uint32_t *table; uint32_t *totals; };
int buckets_for(int64_t n) { return (int)((n + kBucketMean - 1) / kBucketMean); }   // without the "no positive" bucket
int64_t bucket_ws_bytes(int64_t batch)
{
    const int64_t n = batch < kPiece ? batch : kPiece;
    // (+ RSX_MAX_CHUNKS: the chunked layout rounds the bucket count up to a multiple of the number of ranges)
    const int64_t NB = buckets_for(n) + 1 + RSX_MAX_CHUNKS, nblk = (n + kChunk - 1) / kChunk;
    return align256(n * 8) + align256(NB * nblk * 4) + align256(NB * 4 * kTotalStride);
}
BucketWs bucket_carve(void *ws, int64_t n, int64_t NB)
{
    const int64_t nblk = (n + kChunk - 1) / kChunk;
    BucketWs w;
    char *p = (char *)ws;
    w.pairs = (uint2 *)p;      p += align256(n * 8);
    w.table = (uint32_t *)p;   p += align256(NB * nblk * 4);
    w.totals = (uint32_t *)p;
    return w;
}

RSX_API int64_t rsx_bpr_sample_workspace(int64_t batch, int64_t num_items)
{
    if (batch < 0 || num_items <= 0 || num_items >= (1ll << 31)) return RSX_E_INVALID;
    if (batch == 0) return 0;
    const int64_t sort_bytes = 4 * align256(batch * 4) + align256((int64_t)sort_temp_bytes(batch, key_bits(num_items)));
    const int64_t bucket_bytes = bucket_ws_bytes(batch);
    return sort_bytes > bucket_bytes ? sort_bytes : bucket_bytes;
}

RSX_API int64_t rsx_bpr_item_cdf_workspace(int64_t num_items)
{
    return num_items > 0 ? align256(num_items * 8) : RSX_E_INVALID;
}

RSX_API int rsx_bpr_build_item_cdf(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                                   int64_t num_items, uint32_t *cdf_out, void *ws, int64_t ws_bytes,
                                   rsx_stream_t stream)
{
    RSX_CHECK_ARG(indptr_dev && indices_dev && cdf_out && ws, "null pointer");
    RSX_CHECK_ARG(num_users > 0 && num_items > 0 && num_items < (1ll << 31), "bad shape");
    RSX_CHECK_ARG(ws_bytes >= rsx_bpr_item_cdf_workspace(num_items), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *mass = (unsigned long long *)ws;
    if (hipMemsetAsync(mass, 0, (size_t)num_items * 8, st) != hipSuccess) {
        rsx_set_error("rsx_bpr_build_item_cdf: memset failed");
        return RSX_E_HIP;
    }
    hipLaunchKernelGGL(item_mass_kernel, dim3(grid_1d(num_users)), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                       num_users, num_items, mass);
    hipLaunchKernelGGL(item_cdf_kernel, dim3(1), dim3(kScanBlock), 0, st, mass, num_items, cdf_out);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}


RSX_API int rsx_bpr_sample(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                           int64_t num_items, int64_t batch, uint64_t seed, uint64_t step,
                           int64_t epoch_pos, int neg_block, uint64_t neg_key, unsigned flags,
                           void *ws, int64_t ws_bytes, const uint64_t *user_sig_dev,
                           const uint32_t *item_cdf_dev, int32_t *u_out, int32_t *i_out, int32_t *j_out,
                           rsx_stream_t stream)
{
    RSX_CHECK_ARG(indptr_dev && indices_dev && u_out && i_out && j_out, "null pointer");
    RSX_CHECK_ARG(num_users > 0 && num_users < (1ll << 31) && num_items > 0 && num_items < (1ll << 31),
                  "table sizes must fit int32");
    RSX_CHECK_ARG(batch >= 0 && epoch_pos >= 0, "negative size");
    RSX_CHECK_ARG(neg_block >= 0 && neg_block <= kMaxNegBlock, "neg_block must be in [0, 16]");
    if (batch == 0) return RSX_OK;
    const int hb = (batch == num_users && epoch_pos % num_users == 0) ? -1 : half_bits_for(num_users);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & RSX_SAMPLE_SORT_POS)) {
        hipLaunchKernelGGL(bpr_sample_kernel, dim3(grid_1d(batch)), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                           num_users, num_items, batch, seed, step, epoch_pos, hb, neg_block, neg_key, u_out,
                           i_out, j_out);
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    const int64_t need = rsx_bpr_sample_workspace(batch, num_items);
    if (ws == nullptr || ws_bytes < need) {
        rsx_set_error("rsx_bpr_sample: RSX_SAMPLE_SORT_POS needs a workspace of %lld bytes, got %lld",
                      (long long)need, (long long)ws_bytes);
        return RSX_E_WORKSPACE;
    }
    if (item_cdf_dev != nullptr) {
        for (int64_t piece_lo = 0; piece_lo < batch; piece_lo += kPiece) {
            const int64_t n = (batch - piece_lo < kPiece) ? batch - piece_lo : kPiece;
            const int nbm = buckets_for(n), NB = nbm + 1;
            const int nblk = (int)((n + kChunk - 1) / kChunk);
            const BucketWs w = bucket_carve(ws, n, NB);
            const ChunkArgs ca{1, 0, ChunkGeom{}, nullptr, false};
            if (!RSX_SAMPLE_ABL_HOST(1) && hipMemsetAsync(w.totals, 0, (size_t)NB * 4 * kTotalStride, st) != hipSuccess) {
                rsx_set_error("rsx_bpr_sample: memset failed");
                return RSX_E_HIP;
            }
            const size_t lds = ((size_t)((NB + 3) & ~3)) * 4 + (size_t)kChunk * sizeof(uint2);
            if (!RSX_SAMPLE_ABL_HOST(1))
            hipLaunchKernelGGL(bucket_chunk_kernel, dim3(nblk), dim3(kChunkThreads), lds, st, indptr_dev, indices_dev,
                               item_cdf_dev, num_users, num_items, piece_lo, n, seed, step, epoch_pos, hb, nbm, nblk,
                               w.pairs, w.table, w.totals, ca);
            if (!RSX_SAMPLE_ABL_HOST(2))
            hipLaunchKernelGGL(bucket_sort_kernel, dim3(NB), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                               user_sig_dev, num_items, batch, piece_lo, seed, step, neg_block, neg_key, nbm, nblk,
                               lds_sort_cap(), w.pairs, w.table, w.totals, u_out, i_out, j_out, ca);
        }
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    const int64_t arr = align256(batch * 4);
    uint32_t *keys_in = (uint32_t *)ws;
    uint32_t *vals_in = (uint32_t *)((char *)ws + arr);
    uint32_t *keys_out = (uint32_t *)((char *)ws + 2 * arr);
    uint32_t *vals_out = (uint32_t *)((char *)ws + 3 * arr);
    void *temp = (char *)ws + 4 * arr;
    const int bits = key_bits(num_items);
    size_t temp_bytes = sort_temp_bytes(batch, bits);
    hipLaunchKernelGGL(sample_ui_kernel, dim3(grid_1d(batch)), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                       num_users, num_items, batch, seed, step, epoch_pos, hb, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs<sort_config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                             (size_t)batch, 0, (unsigned)bits, st);
    if (e != hipSuccess) {
        rsx_set_error("rsx_bpr_sample: radix sort failed: %s", hipGetErrorString(e));
        return RSX_E_HIP;
    }
    hipLaunchKernelGGL(sample_neg_kernel, dim3(grid_1d(batch)), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                       num_items, batch, seed, step, neg_block, neg_key, keys_out, vals_out, user_sig_dev, u_out, i_out, j_out);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int64_t rsx_chunk_rows(int64_t items_real, int chunks, int neg_block)
{
    if (items_real <= 0 || chunks < 1 || chunks > RSX_MAX_CHUNKS || neg_block < 0 || neg_block > kMaxNegBlock) return RSX_E_INVALID;
    return chunk_geom(items_real, chunks, neg_block < 1 ? 1 : neg_block).Ic;
}

RSX_API int rsx_bpr_sample_chunked(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                                   int64_t num_items, int64_t items_real, int chunks, int64_t batch, uint64_t seed,
                                   uint64_t step, int64_t epoch_pos, int neg_block, uint64_t neg_key, void *ws,
                                   int64_t ws_bytes, const uint64_t *user_sig_dev, const uint32_t *item_cdf_dev,
                                   int32_t *u_out, int32_t *i_out, int32_t *j_out, int64_t *chunk_pos_out,
                                   rsx_stream_t stream)
{
    RSX_CHECK_ARG(indptr_dev && indices_dev && u_out && i_out && j_out && chunk_pos_out && item_cdf_dev, "null pointer");
    RSX_CHECK_ARG(chunks >= 2 && chunks <= RSX_MAX_CHUNKS, "chunks must be in [2, RSX_MAX_CHUNKS]");
    RSX_CHECK_ARG(neg_block >= 0 && neg_block <= kMaxNegBlock, "neg_block must be in [0, 16]");
    RSX_CHECK_ARG(num_users > 0 && num_users < (1ll << 31) && items_real > 0 && num_items < (1ll << 31), "table sizes must fit int32");
    // neg_block = 0: no item blocks -- the negative of a position is uniform over the real items of its range (batches below two
    // triplets per item, where the blocked layout has nothing to sum on chip); the ranges are then ceil(items_real / chunks) rows
    const bool whole = neg_block == 0;
    const ChunkGeom g = chunk_geom(items_real, chunks, whole ? 1 : neg_block);
    RSX_CHECK_ARG(num_items == g.Ic * chunks, "num_items must be chunks * rsx_chunk_rows(items_real, chunks, neg_block)");
    RSX_CHECK_ARG(batch >= 0 && batch <= kPiece && epoch_pos >= 0, "the chunked layout orders at most 2^21 positions");
    if (batch == 0) return RSX_OK;
    const int64_t need = rsx_bpr_sample_workspace(batch, num_items);
    if (ws == nullptr || ws_bytes < need) {
        rsx_set_error("rsx_bpr_sample_chunked: needs a workspace of %lld bytes, got %lld", (long long)need, (long long)ws_bytes);
        return RSX_E_WORKSPACE;
    }
    const int hb = (batch == num_users && epoch_pos % num_users == 0) ? -1 : half_bits_for(num_users);
    hipStream_t st = (hipStream_t)stream;
    const int NBc = (buckets_for(batch) + chunks - 1) / chunks;
    const int nbm = NBc * chunks, NB = nbm + 1;
    const int nblk = (int)((batch + kChunk - 1) / kChunk);
    const BucketWs w = bucket_carve(ws, batch, NB);
    if (hipMemsetAsync(w.totals, 0, (size_t)NB * 4 * kTotalStride, st) != hipSuccess) {
        rsx_set_error("rsx_bpr_sample_chunked: memset failed");
        return RSX_E_HIP;
    }
    const ChunkArgs ca{chunks, NBc, g, chunk_pos_out, whole};
    const size_t lds = ((size_t)((NB + 3) & ~3)) * 4 + (size_t)kChunk * sizeof(uint2);
    hipLaunchKernelGGL(bucket_chunk_kernel, dim3(nblk), dim3(kChunkThreads), lds, st, indptr_dev, indices_dev, item_cdf_dev,
                       num_users, num_items, (int64_t)0, batch, seed, step, epoch_pos, hb, nbm, nblk, w.pairs, w.table, w.totals, ca);
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(NB), dim3(kBlock), 0, st, indptr_dev, indices_dev, user_sig_dev, num_items,
                       batch, (int64_t)0, seed, step, neg_block, neg_key, nbm, nblk, lds_sort_cap(), w.pairs, w.table, w.totals,
                       u_out, i_out, j_out, ca);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

// ---- the CSC walk: host side ----------------------------------------------------------------------
struct rsx_csc {
    CscArgs a{};
    int rd_bits = 16;
    int64_t num_users = 0;
    const int64_t *indptr = nullptr;       // the CSR it was built from (identity check only)
    const int32_t *indices = nullptr;
};

namespace {
struct CscLayout { int64_t ntiles, nnz_pad, off_ptr, off_tile, off_users, off_rd, bytes; };
CscLayout csc_layout(int64_t nnz, int64_t num_items)
{
    CscLayout L;
    L.ntiles = nnz > 0 ? (nnz + kCscTile - 1) / kCscTile : 1;
    L.nnz_pad = L.ntiles * kCscTile;
    L.off_ptr = 0;
    L.off_tile = L.off_ptr + align256((num_items + 1) * 8);
    L.off_users = L.off_tile + align256((L.ntiles + 1) * 4);
    L.off_rd = L.off_users + align256(L.nnz_pad * 4);
    L.bytes = L.off_rd + align256(L.nnz_pad * 4);                 // (room for the 32-bit form; the 16-bit form uses half)
    return L;
}
size_t csc_sort_temp_bytes(int64_t nnz, int bits)
{
    size_t n = 0;
    uint32_t *k = nullptr;
    uint64_t *v = nullptr;
    (void)rocprim::radix_sort_pairs<sort_config>(nullptr, n, k, k, v, v, (size_t)(nnz > 0 ? nnz : 1), 0, (unsigned)bits, (hipStream_t)0);
    return n;
}
}  // namespace

RSX_API int64_t rsx_bpr_csc_bytes(int64_t nnz, int64_t num_items)
{
    if (nnz < 0 || nnz >= (1ll << 32) || num_items <= 0 || num_items >= (1ll << 31)) return RSX_E_INVALID;
    return csc_layout(nnz, num_items).bytes;
}

RSX_API int64_t rsx_bpr_csc_workspace(int64_t nnz, int64_t num_items)
{
    if (nnz < 0 || nnz >= (1ll << 32) || num_items <= 0 || num_items >= (1ll << 31)) return RSX_E_INVALID;
    const int64_t n = nnz > 0 ? nnz : 1;
    return 256 + align256(n * 4) + 2 * align256(n * 8) + align256((int64_t)csc_sort_temp_bytes(nnz, key_bits(num_items)));
}

RSX_API int rsx_bpr_build_csc(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users, int64_t num_items, int64_t nnz,
                              void *blob, int64_t blob_bytes, void *ws, int64_t ws_bytes, rsx_stream_t stream, rsx_csc **out)
{
    RSX_CHECK_ARG(indptr_dev && blob && ws && out, "null pointer");
    RSX_CHECK_ARG(num_users > 0 && num_users < (1ll << 31) && num_items > 0 && num_items < (1ll << 31) && nnz >= 0 && nnz < (1ll << 32),
                  "table sizes must fit int32 (nnz: uint32)");
    RSX_CHECK_ARG(indices_dev != nullptr || nnz == 0, "null pointer (indices may be null only when there is no interaction at all)");
    RSX_CHECK_ARG(blob_bytes >= rsx_bpr_csc_bytes(nnz, num_items), "blob smaller than rsx_bpr_csc_bytes(nnz, num_items)");
    RSX_CHECK_ARG(ws_bytes >= rsx_bpr_csc_workspace(nnz, num_items), "workspace smaller than rsx_bpr_csc_workspace(nnz, num_items)");
    hipStream_t st = (hipStream_t)stream;
    const CscLayout L = csc_layout(nnz, num_items);
    const int64_t n = nnz > 0 ? nnz : 1;
    char *w = (char *)ws;
    unsigned long long *max_deg = (unsigned long long *)w;          w += 256;
    uint32_t *keys_out = (uint32_t *)w;                             w += align256(n * 4);
    uint64_t *vals_in = (uint64_t *)w;                              w += align256(n * 8);
    uint64_t *vals_out = (uint64_t *)w;                             w += align256(n * 8);
    void *temp = w;
    // the longest row decides the entry format (this is a set-up call: it waits for the device once)
    unsigned long long mx = 0;
    if (hipMemsetAsync(max_deg, 0, 8, st) != hipSuccess) { rsx_set_error("rsx_bpr_build_csc: memset failed"); return RSX_E_HIP; }
    hipLaunchKernelGGL(csc_entries_kernel, dim3(grid_1d(num_users)), dim3(kBlock), 0, st, indptr_dev, num_users, 0, (uint64_t *)nullptr, max_deg);
    hipError_t e = hipMemcpyAsync(&mx, max_deg, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { rsx_set_error("rsx_bpr_build_csc: reading the longest row failed: %s", hipGetErrorString(e)); return RSX_E_HIP; }
    if (mx > 65535ull) {
        rsx_set_error("rsx_bpr_build_csc: a user holds %llu items; the CSC walk takes rows of at most 65535 (use the bucket sampler)", mx);
        return RSX_E_INVALID;
    }
    const int rd_bits = mx <= 255ull ? 16 : 32;
    const int shift = rd_bits == 16 ? 8 : 16;
    int64_t *ptr = (int64_t *)((char *)blob + L.off_ptr);
    int32_t *tile_item = (int32_t *)((char *)blob + L.off_tile);
    uint32_t *users = (uint32_t *)((char *)blob + L.off_users);
    void *rd = (char *)blob + L.off_rd;
    if (nnz > 0) {
        hipLaunchKernelGGL(csc_entries_kernel, dim3(grid_1d(num_users)), dim3(kBlock), 0, st, indptr_dev, num_users, shift, vals_in, max_deg);
        size_t temp_bytes = csc_sort_temp_bytes(nnz, key_bits(num_items));
        // stable: inside an item the users stay ascending (the CSR is walked user by user)
        e = rocprim::radix_sort_pairs<sort_config>(temp, temp_bytes, reinterpret_cast<const uint32_t *>(indices_dev), keys_out, vals_in, vals_out,
                                                   (size_t)nnz, 0, (unsigned)key_bits(num_items), st);
        if (e != hipSuccess) { rsx_set_error("rsx_bpr_build_csc: radix sort failed: %s", hipGetErrorString(e)); return RSX_E_HIP; }
    }
    if (rd_bits == 16)
        hipLaunchKernelGGL(csc_split_kernel<uint16_t>, dim3(grid_1d(L.nnz_pad)), dim3(kBlock), 0, st, keys_out, vals_out, nnz, L.nnz_pad, num_items,
                           users, (uint16_t *)rd, ptr, tile_item, (int)L.ntiles);
    else
        hipLaunchKernelGGL(csc_split_kernel<uint32_t>, dim3(grid_1d(L.nnz_pad)), dim3(kBlock), 0, st, keys_out, vals_out, nnz, L.nnz_pad, num_items,
                           users, (uint32_t *)rd, ptr, tile_item, (int)L.ntiles);
    RSX_CHECK_LAUNCH();
    rsx_csc *c = new (std::nothrow) rsx_csc();
    if (c == nullptr) { rsx_set_error("rsx_bpr_build_csc: out of memory"); return RSX_E_INVALID; }
    c->a = CscArgs{ptr, tile_item, users, rd, nnz, num_items, (int)L.ntiles};
    c->rd_bits = rd_bits; c->num_users = num_users; c->indptr = indptr_dev; c->indices = indices_dev;
    *out = c;
    return RSX_OK;
}

RSX_API void rsx_csc_destroy(rsx_csc *c) { delete c; }

RSX_API int rsx_csc_info(const rsx_csc *c, int64_t *nnz, int64_t *num_items, int *entry_bytes, int64_t *tiles)
{
    RSX_CHECK_ARG(c != nullptr, "null csc");
    if (nnz) *nnz = c->a.nnz;
    if (num_items) *num_items = c->a.I;
    if (entry_bytes) *entry_bytes = 4 + c->rd_bits / 8;
    if (tiles) *tiles = c->a.ntiles;
    return RSX_OK;
}

namespace {
struct CscWs { int64_t *n_live; uint32_t *border_tile, *border_rank, *counts, *first_pos; uint2 *stage; int64_t bytes; };
CscWs csc_ws_carve(void *ws, int64_t nnz)
{
    const int64_t ntiles = nnz > 0 ? (nnz + kCscTile - 1) / kCscTile : 1;
    char *p = (char *)ws;
    CscWs w;
    w.n_live = (int64_t *)p;
    w.border_tile = (uint32_t *)(p + 64);
    w.border_rank = (uint32_t *)(p + 128);
    int64_t off = 256;
    w.counts = (uint32_t *)(p + off);     off += align256(ntiles * 4);
    w.first_pos = (uint32_t *)(p + off);  off += align256((ntiles + 1) * 4);
    w.stage = (uint2 *)(p + off);         off += ntiles * kCscTile * 8;      // a slice per tile with room for EVERY entry of the tile
    w.bytes = off;
    return w;
}
static_assert(RSX_MAX_CHUNKS * 4 <= 64, "the ranges' border words sit in two 64-byte slots of the workspace head");
}  // namespace

RSX_API int64_t rsx_bpr_sample_csc_workspace(int64_t nnz)
{
    if (nnz < 0 || nnz >= (1ll << 32)) return RSX_E_INVALID;
    return csc_ws_carve(nullptr, nnz).bytes;
}

// (internal, rsx_common.h) the native loop asks before it takes the CSC walk
bool rsx_csc_matches(const rsx_csc *c, const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users, int64_t num_items)
{
    return c != nullptr && c->indptr == indptr_dev && c->indices == indices_dev && c->num_users == num_users && c->a.I == num_items;
}

RSX_API int rsx_bpr_sample_csc(const rsx_csc *csc, const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                               int64_t num_items, int64_t items_real, int chunks, uint64_t seed, uint64_t step, int neg_block,
                               uint64_t neg_key, void *ws, int64_t ws_bytes, const uint64_t *user_sig_dev, int32_t *u_out,
                               int32_t *i_out, int32_t *j_out, int64_t *chunk_pos_out, rsx_stream_t stream)
{
    RSX_CHECK_ARG(csc && indptr_dev && (indices_dev || csc->a.nnz == 0) && u_out && i_out && j_out && ws, "null pointer");
    RSX_CHECK_ARG(rsx_csc_matches(csc, indptr_dev, indices_dev, num_users, num_items),
                  "the CSC was built from another CSR (rsx_bpr_build_csc takes the SAME indptr, indices, num_users, num_items)");
    RSX_CHECK_ARG(neg_block >= 0 && neg_block <= kMaxNegBlock, "neg_block must be in [0, 16]");
    RSX_CHECK_ARG(chunks >= 0 && chunks <= RSX_MAX_CHUNKS && (chunks <= 1 || chunk_pos_out != nullptr), "chunks must be in [0, RSX_MAX_CHUNKS]; chunks > 1 needs chunk_pos_out");
    RSX_CHECK_ARG(ws_bytes >= rsx_bpr_sample_csc_workspace(csc->a.nnz), "workspace smaller than rsx_bpr_sample_csc_workspace(nnz)");
    const bool chunked_layout = chunks > 1;
    const bool whole = chunked_layout && neg_block == 0;
    ChunkGeom g{};
    if (chunked_layout) {
        RSX_CHECK_ARG(items_real > 0, "chunks > 1 needs items_real");
        g = chunk_geom(items_real, chunks, whole ? 1 : neg_block);
        RSX_CHECK_ARG(num_items == g.Ic * chunks, "num_items must be chunks * rsx_chunk_rows(items_real, chunks, neg_block)");
    }
    const int64_t B = num_users;                                   // the batch IS one pass over the users
    hipStream_t st = (hipStream_t)stream;
    const CscWs w = csc_ws_carve(ws, csc->a.nnz);
    const uint64_t k = csc_step_key(seed, step);
    const uint32_t k0 = (uint32_t)k, k1 = (uint32_t)(k >> 32);
    const ChunkArgs ca{chunked_layout ? chunks : 1, 0, g, chunked_layout ? chunk_pos_out : nullptr, whole};
    // select (every tile on its own) -> scan of the tiles' counts (one workgroup) -> place -> negatives per ordered position
    if (csc->rd_bits == 16)
        hipLaunchKernelGGL(csc_select_kernel<uint16_t>, dim3(csc->a.ntiles), dim3(kCscThreads), 0, st, csc->a, k0, k1, w.stage, w.counts, ca.C, g.Ic,
                           w.border_tile, w.border_rank);
    else
        hipLaunchKernelGGL(csc_select_kernel<uint32_t>, dim3(csc->a.ntiles), dim3(kCscThreads), 0, st, csc->a, k0, k1, w.stage, w.counts, ca.C, g.Ic,
                           w.border_tile, w.border_rank);
    hipLaunchKernelGGL(csc_scan_kernel, dim3(1), dim3(kCscScanThreads), 0, st, w.counts, w.first_pos, csc->a.ntiles, w.n_live);
    hipLaunchKernelGGL(csc_place_kernel, dim3(csc->a.ntiles), dim3(kCscThreads), 0, st, w.stage, w.counts, w.first_pos, csc->a.ntiles, u_out, i_out,
                       ca.C, w.border_tile, w.border_rank, ca.chunk_pos_out);
    int64_t *n_live = w.n_live;
    const int64_t blocks = ceil_div64(B, kCscNegBlock) + ca.C + 1;
    hipLaunchKernelGGL(csc_neg_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, indptr_dev, indices_dev, user_sig_dev, num_items, B, seed, step,
                       neg_block, neg_key, n_live, u_out, i_out, j_out, ca);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}
