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
//   sorted  (RSX_SAMPLE_SORT_POS) the (i, u) pairs are radix-sorted by positive item,
//           then negatives are drawn per sorted position from a keyed item block.
//           Equal positives become contiguous, so the step kernel sums their gradient
//           in registers, and all negatives of an item block belong to one wavefront.
// The sort is rocPRIM's device radix sort (a plain library primitive, like a library
// GEMM); everything arithmetic stays in this repository's kernels.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "rsx_common.h"

namespace {

constexpr int kBlock = 256;

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

// 16-bit-key variant (catalogs below 2^17 items): rocPRIM sorts 2-byte keys with Onesweep,
// measured ~2.5x faster than the 4-byte merge-sort path.  key = item >> shift (shift <= 1), the
// value carries (user, item); the step kernel keeps one run accumulator per item parity.
__global__ __launch_bounds__(kBlock) void sample_ui16_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t U, int64_t I,
    int64_t B, uint64_t seed, uint64_t step, int64_t epoch_pos, int hb, int shift,
    uint16_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        const uint32_t u = user_at(epoch_pos + b, U, hb, seed);
        uint32_t s = rng_seed(seed, step, (uint64_t)b, 0);
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        const uint32_t deg = (uint32_t)(hi - lo);
        uint32_t item = (uint32_t)I;                       // "no positive": sorts last
        if (deg > 0 && (int64_t)deg < I) item = (uint32_t)indices[lo + (int64_t)(((uint64_t)xorshift32(s) * deg) >> 32)];
        keys[b] = (uint16_t)(item >> shift);
        vals[b] = (u << shift) | (item & ((1u << shift) - 1u));
    }
}

__global__ __launch_bounds__(kBlock) void sample_neg16_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t I, int64_t B,
    uint64_t seed, uint64_t step, int neg_block, uint64_t neg_key, int shift,
    const uint16_t *__restrict__ keys_sorted, const uint32_t *__restrict__ vals_sorted,
    const uint64_t *__restrict__ user_sig,
    int32_t *__restrict__ u_out, int32_t *__restrict__ i_out, int32_t *__restrict__ j_out)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < B; p += (int64_t)gridDim.x * kBlock) {
        const uint32_t v = vals_sorted[p];
        const uint32_t u = v >> shift;
        const uint32_t item = ((uint32_t)keys_sorted[p] << shift) | (v & ((1u << shift) - 1u));
        int32_t pi = -1, nj = -1;
        if ((int64_t)item < I) {
            uint32_t s = rng_seed(seed, step, (uint64_t)p, 0x5bd1e995ull);
            pi = (int32_t)item;
            bool done = false;
            if (user_sig != nullptr && neg_block > 0) {
                // one 8-byte read instead of indptr + the row: the user's signature has a bit for
                // every item block holding one of its positives; a clear bit proves the whole block
                // negative for this user (73 % of the draws at 20 positives per user)
                int64_t neg_lo, neg_n;
                neg_range(I, p, B, neg_block, neg_key, neg_lo, neg_n);
                if (((user_sig[u] >> sig_bit(neg_lo / neg_block)) & 1ull) == 0ull) {
                    nj = (int32_t)(neg_lo + (int64_t)(((uint64_t)xorshift32(s) * (uint64_t)neg_n) >> 32));
                    done = true;
                }
            }
            if (!done) nj = draw_negative(indices, indptr[u], indptr[u + 1], I, p, B, neg_block, neg_key, s);
        }
        u_out[p] = (int32_t)u; i_out[p] = pi; j_out[p] = nj;
    }
}

// user_sig[u] = OR over the user's positives of (1 << sig_bit(item / neg_block)); static per CSR
__global__ __launch_bounds__(kBlock) void build_signature_kernel(const int64_t *__restrict__ indptr,
                                                                 const int32_t *__restrict__ indices, int64_t U,
                                                                 int neg_block, uint64_t *__restrict__ sig)
{
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < U; u += (int64_t)gridDim.x * kBlock) {
        uint64_t m = 0ull;
        for (int64_t q = indptr[u]; q < indptr[u + 1]; ++q) m |= 1ull << sig_bit(indices[q] / neg_block);
        sig[u] = m;
    }
}

// sorted layout, pass 2: negatives per SORTED position p
__global__ __launch_bounds__(kBlock) void sample_neg_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t I, int64_t B,
    uint64_t seed, uint64_t step, int neg_block, uint64_t neg_key,
    const uint32_t *__restrict__ keys_sorted, const uint32_t *__restrict__ vals_sorted,
    int32_t *__restrict__ u_out, int32_t *__restrict__ i_out, int32_t *__restrict__ j_out)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < B; p += (int64_t)gridDim.x * kBlock) {
        const uint32_t u = vals_sorted[p];
        const uint32_t key = keys_sorted[p];
        int32_t pi = -1, nj = -1;
        if ((int64_t)key < I) {
            uint32_t s = rng_seed(seed, step, (uint64_t)p, 0x5bd1e995ull);
            pi = (int32_t)key;
            nj = draw_negative(indices, indptr[u], indptr[u + 1], I, p, B, neg_block, neg_key, s);
        }
        u_out[p] = (int32_t)u; i_out[p] = pi; j_out[p] = nj;
    }
}

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

bool use_key16(int64_t num_items) { return num_items < (1ll << 17) - 1; }   // key 0xFFFF>>... reserved for "no positive"

size_t sort16_temp_bytes(int64_t batch)
{
    size_t n = 0;
    uint16_t *k = nullptr;
    uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, n, k, k, v, v, (size_t)batch, 0, 16u, (hipStream_t)0);
    return n;
}

}  // namespace

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

RSX_API int64_t rsx_bpr_sample_workspace(int64_t batch, int64_t num_items)
{
    if (batch < 0 || num_items <= 0 || num_items >= (1ll << 31)) return RSX_E_INVALID;
    if (batch == 0) return 0;
    if (use_key16(num_items))
        return 2 * align256(batch * 2) + 2 * align256(batch * 4) + align256((int64_t)sort16_temp_bytes(batch));
    return 4 * align256(batch * 4) + align256((int64_t)sort_temp_bytes(batch, key_bits(num_items)));
}

RSX_API int rsx_bpr_sample(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                           int64_t num_items, int64_t batch, uint64_t seed, uint64_t step,
                           int64_t epoch_pos, int neg_block, uint64_t neg_key, unsigned flags,
                           void *ws, int64_t ws_bytes, const uint64_t *user_sig_dev, int32_t *u_out,
                           int32_t *i_out, int32_t *j_out, rsx_stream_t stream)
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
    if (use_key16(num_items)) {
        const int shift = num_items < (1ll << 16) ? 0 : 1;
        const int64_t ka = align256(batch * 2), va = align256(batch * 4);
        uint16_t *k_in = (uint16_t *)ws, *k_out = (uint16_t *)((char *)ws + ka);
        uint32_t *v_in = (uint32_t *)((char *)ws + 2 * ka), *v_out = (uint32_t *)((char *)ws + 2 * ka + va);
        void *tmp = (char *)ws + 2 * ka + 2 * va;
        size_t tmp_bytes = sort16_temp_bytes(batch);
        hipLaunchKernelGGL(sample_ui16_kernel, dim3(grid_1d(batch)), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                           num_users, num_items, batch, seed, step, epoch_pos, hb, shift, k_in, v_in);
        hipError_t e16 = rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)batch, 0, 16u, st);
        if (e16 != hipSuccess) {
            rsx_set_error("rsx_bpr_sample: radix sort failed: %s", hipGetErrorString(e16));
            return RSX_E_HIP;
        }
        hipLaunchKernelGGL(sample_neg16_kernel, dim3(grid_1d(batch)), dim3(kBlock), 0, st, indptr_dev, indices_dev,
                           num_items, batch, seed, step, neg_block, neg_key, shift, k_out, v_out, user_sig_dev,
                           u_out, i_out, j_out);
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
                       num_items, batch, seed, step, neg_block, neg_key, keys_out, vals_out, u_out, i_out, j_out);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}
