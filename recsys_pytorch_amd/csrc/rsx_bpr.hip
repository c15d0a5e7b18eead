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
// d fp32 is spread over a lane group of 32 lanes (d/32 floats per lane); a 64-lane
// wavefront owns two triplets at once, the dot product is a butterfly inside the
// lane group, and every load/store/atomic instruction covers whole 128-byte lines
// of a row (see ROW LAYOUT below).
#include <math.h>

#include "rsx_common.h"

namespace {

constexpr int kBlock = 256;           // 4 wavefronts per workgroup
constexpr int kWavesPerBlock = kBlock / 64;

// ROW LAYOUT.  A row of D floats is spread over the 32 lanes of a lane group; lane k holds
// elements k, k+32, ... (EPL = D/32 of them).  Every wave instruction therefore covers ONE whole
// 128-byte line per row, for loads, stores and atomics alike.  (The obvious alternative, one
// dwordx4 per lane = lane k holds [4k,4k+4), makes each atomic instruction touch four lines per
// row and measured 3.7x slower at d=128: 226 us vs 60.7 us for B=65536, tools/microbench.py.)
constexpr int LPR = 32;   // lanes per row
constexpr int TPW = 2;    // lane groups (triplets) per wavefront

//
// ADDRESSING.  A row is named by a byte offset from its table's base, OffT wide.  OffT = uint32_t
// (tables below 4 GB, every BASELINE shape per GPU): the base stays in SGPRs and each row costs ONE
// VGPR and one 32-bit multiply-add -- global_load_dword v, v_off, s[base:base+1] offset:c*128 --
// instead of a 64-bit pointer pair and 64-bit VALU address math per access (the step kernels were
// spilling registers into scratch inside the triplet loop, and a spill reload is a VMEM operation
// that waits behind the previous trip's stores and atomics).  OffT = uint64_t: same code, any size.
template <int D, typename OffT>
__device__ __forceinline__ OffT row_off(int32_t row, int k)
{
    return (OffT)(uint32_t)row * (OffT)(D * 4) + (OffT)(k * 4);
}
template <typename OffT>
__device__ __forceinline__ float *at(const float *base, OffT off)
{
    // (pointer arithmetic, not an integer round trip: the address must stay provably global memory)
    return reinterpret_cast<float *>(reinterpret_cast<char *>(const_cast<float *>(base)) + off);
}

template <int D>
struct Row {
    static constexpr int EPL = D / 32;
    float v[EPL];
    static __device__ __forceinline__ int elem(int k, int c) { return k + 32 * c; }
    template <typename OffT>
    __device__ __forceinline__ void load_at(const float *base, OffT off)
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) v[c] = at(base, off)[32 * c];
    }
    template <typename OffT>
    __device__ __forceinline__ void store_at(float *base, OffT off) const
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) at(base, off)[32 * c] = v[c];
    }
    // streaming forms (`nt`) for rows a step touches exactly once -- the user rows of a unique-user batch: they
    // are kept at the lowest retention in the XCD's 4 MB L2, which leaves it to the item rows that ARE read again
    // (round 2, same box: blocked kernel in the loop 342-355 -> 322 us, 2.50-2.61 -> 2.71e9 triplets/s; with `nt`
    // on only the loads or only the stores: no gain; on the index loads or in the apply sweep: slightly slower)
    template <typename OffT>
    __device__ __forceinline__ void load_once_at(const float *base, OffT off)
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) v[c] = __builtin_nontemporal_load(at(base, off) + 32 * c);
    }
    template <typename OffT>
    __device__ __forceinline__ void store_once_at(float *base, OffT off) const
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) __builtin_nontemporal_store(v[c], at(base, off) + 32 * c);
    }
    template <typename OffT>
    __device__ __forceinline__ void atomic_axpy_at(float *base, OffT off, float s) const
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) rsx_atomic_add(at(base, off) + 32 * c, s * v[c]);
    }
    __device__ __forceinline__ void load(const float *row, int k)
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) v[c] = row[elem(k, c)];
    }
    __device__ __forceinline__ void store(float *row, int k) const
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) row[elem(k, c)] = v[c];
    }
    // row[...] += s * v   (hardware fp32 atomics, no return)
    __device__ __forceinline__ void atomic_axpy(float *row, int k, float s) const
    {
#pragma unroll
        for (int c = 0; c < EPL; ++c) rsx_atomic_add(row + elem(k, c), s * v[c]);
    }
};

// sum over the 32 lanes of a lane group, result in every lane of the group.  Pure cross-lane
// VALU (DPP inside a 16-lane row, v_permlane16/32_swap across rows): no LDS round trips
// (__shfl_xor lowers to ds_bpermute_b32, 10 of them per triplet on the dependent path).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float group_sum(float x)
{
    x += dpp_mov<0xB1>(x);                       // quad_perm [1,0,3,2]  : lane ^ 1
    x += dpp_mov<0x4E>(x);                       // quad_perm [2,3,0,1]  : lane ^ 2
    x += dpp_mov<0x141>(x);                      // row_half_mirror: the other quad of the 8
    x += dpp_mov<0x140>(x);                      // row_mirror     : the other half of the 16
    const unsigned b = __builtin_bit_cast(unsigned, x);      // rows 0<->1, 2<->3
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// sigmoid(-x) (the coefficient of the gradient, MF.py:105 differentiated) and softplus(-x) = -log(sigmoid(x)) (the loss; stable for
// any x: the reference's sigmoid().log() underflows for x < -88, same value elsewhere) from ONE exponential e = exp(-|x|) in (0, 1]:
//     sigmoid(-x) = (x >= 0 ? e : 1) / (1 + e)        softplus(-x) = max(-x, 0) + log(1 + e)
// on the transcendental unit (v_exp_f32, v_rcp_f32, v_log_f32: 1 ulp each) -- a dozen vector instructions per triplet.  Round 4: the
// C library's log1pf (a double-float routine, ~140 vector instructions) plus an IEEE division were 60 % of the vector instructions
// of the step kernels' trip, and the kernels had become issue-bound (SQ_ACTIVE_INST_VALU: the vector pipes ~80 % busy at the
// headline).  Absolute error of the loss <= 1e-7 per triplet (1 + e rounds e away below 6e-8); relative error of the coefficient 2e-7.
#ifndef RSX_FAST_LOSS
#define RSX_FAST_LOSS 1         // 0: log1pf and the IEEE division (development A/B)
#endif
__device__ __forceinline__ float sigmoid_neg(float x, float &one_plus_e)
{
#if RSX_FAST_LOSS
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * -1.4426950408889634f);
    one_plus_e = 1.0f + e;
    return (x >= 0.0f ? e : 1.0f) * __builtin_amdgcn_rcpf(one_plus_e);
#else
    one_plus_e = 1.0f + __expf(-fabsf(x));
    return 1.0f / (1.0f + __expf(x));
#endif
}
__device__ __forceinline__ float softplus_neg(float x, float one_plus_e)
{
#if RSX_FAST_LOSS
    return fmaxf(-x, 0.0f) + __builtin_amdgcn_logf(one_plus_e) * 0.6931471805599453f;
#else
    return fmaxf(-x, 0.0f) + log1pf(one_plus_e - 1.0f);
#endif
}

// EPL contiguous floats of a wave-private LDS tile (one ds_read/ds_write of 4, 8 or 16 bytes)
template <int EPL>
__device__ __forceinline__ void tile_load(const float *cell, float (&a)[EPL])
{
    if constexpr (EPL == 8) {
        const float4 t = *reinterpret_cast<const float4 *>(cell), w = *reinterpret_cast<const float4 *>(cell + 4);
        a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w; a[4] = w.x; a[5] = w.y; a[6] = w.z; a[7] = w.w;
    }
    else if constexpr (EPL == 4) { const float4 t = *reinterpret_cast<const float4 *>(cell); a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w; }
    else if constexpr (EPL == 2) { const float2 t = *reinterpret_cast<const float2 *>(cell); a[0] = t.x; a[1] = t.y; }
    else { a[0] = *cell; }
}
template <int EPL>
__device__ __forceinline__ void tile_store(float *cell, const float (&a)[EPL])
{
    if constexpr (EPL == 8) {
        *reinterpret_cast<float4 *>(cell) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4 *>(cell + 4) = make_float4(a[4], a[5], a[6], a[7]);
    }
    else if constexpr (EPL == 4) *reinterpret_cast<float4 *>(cell) = make_float4(a[0], a[1], a[2], a[3]);
    else if constexpr (EPL == 2) *reinterpret_cast<float2 *>(cell) = make_float2(a[0], a[1]);
    else *cell = a[0];
}

// Hot item rows (popular items hit by thousands of triplets per step) serialise at the
// memory-side atomic unit (~40 same-line ops/us measured).  Their gradient is therefore
// spread over `replicas` private copies, picked by wavefront id, and folded into G by
// fold_hot_kernel before G is consumed.  slot == nullptr disables the indirection.
struct HotMap {
    const int32_t *slot;   // [num_items] hot slot of an item or -1
    float *ghot;           // [n_hot x replicas x D], zero between steps
    int replicas;          // power of two
    // (the native loop's small batches, plain kernel only) [num_items] bytes, zero between steps: the kernel marks every row of G it adds
    // to, and the apply that follows visits the marked rows only instead of sweeping the table (apply_item_grad_touched_kernel)
    uint8_t *touched = nullptr;
};

// MODE 0: users unique in the batch -> P[u] updated in place by its owner group.
// MODE 1: users may repeat          -> user deltas summed into GU[owner slot].
// MODE 2: gradients only            -> dP summed into the dense buffer GU[u] (P untouched);
//                                      the optimizer sweep (adam_apply_kernel) consumes it.
// PASS (compile time): which side of the step this launch WRITES.  kPassItems = the item sums
// (G / replicas), kPassUsers = the user rows; both = the whole step, neither = loss only
// (RSX_NO_UPDATE).  RSX_ITEMS_ONLY / RSX_USERS_ONLY are the two halves of the two-pass step.
constexpr int kPassItems = 1, kPassUsers = 2, kPassBoth = 3;

template <int D, int MODE, int PASS, typename OffT>
__global__ __launch_bounds__(kBlock) void bpr_step_kernel(
    float *__restrict__ P, const float *__restrict__ Q, float *__restrict__ G,
    const int32_t *__restrict__ U_idx, const int32_t *__restrict__ I_idx,
    const int32_t *__restrict__ J_idx, int64_t B, float lr, float inv_batch,
    float *__restrict__ loss_acc, const int32_t *__restrict__ owner, float *__restrict__ GU,
    HotMap hot)
{
    static_assert(MODE == 0 || PASS == kPassBoth, "the split passes exist for the in-place (unique users) mode only");
    constexpr int EPL = D / 32;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

    float loss_local = 0.0f;
    int64_t b = wave * TPW + sub;
    const int64_t stride = nwaves * TPW;
    // SOFTWARE PIPELINE over the trips, as in the blocked kernel below (see the comment there): the rows of the next triplet are
    // requested at the END of a trip, behind its atomics and stores, and the trip's one wait drains both together.
    // (u, i, j): the triplet being processed; (un, in, jn): the one whose rows are in flight; (um, im, jm): the one after it.
    int32_t u = -1, i = -1, j = -1, un = -1, in = -1, jn = -1, um = -1, im = -1, jm = -1;
    if (b < B) { un = U_idx[b]; in = I_idx[b]; jn = J_idx[b]; }
    Row<D> p, qi, qj;
    int32_t hs = -1;             // replica slot of the positive item (HotMap), requested WITH the rows: fetched after them it is a
                                 // second dependent round trip in every trip
    const bool has_hot = (PASS & kPassItems) != 0 && hot.slot != nullptr;
    bool live_n = (b < B) && (in >= 0);
    auto request_rows = [&]() __attribute__((always_inline)) {
        if (live_n) {
            if constexpr (MODE == 0) p.load_once_at(P, row_off<D, OffT>(un, k)); else p.load_at(P, row_off<D, OffT>(un, k));   // unique users: touched once
            qi.load_at(Q, row_off<D, OffT>(in, k));
            qj.load_at(Q, row_off<D, OffT>(jn, k));
            if (has_hot) hs = hot.slot[in];
        }
    };
    request_rows();
    if (b + stride < B) { um = U_idx[b + stride]; im = I_idx[b + stride]; jm = J_idx[b + stride]; }
    while (b - sub < B) {   // wave-uniform trip count
        const int64_t bn = b + stride;
        u = un; i = in; j = jn; un = um; in = im; jn = jm;      // (first use of the prefetched indices / the rows: the trip's one wait)
        const bool live = live_n;
        if (live) {
            const OffT u_off = row_off<D, OffT>(u, k), i_off = row_off<D, OffT>(i, k), j_off = row_off<D, OffT>(j, k);
            // x = <p, q_i> - <p, q_j> as ONE dot product with the difference row (which the user update needs anyway): one
            // butterfly instead of two.  (inactive groups skip the butterfly; the partners of every cross-lane step are
            //  inside the same 32-lane group, which is live or dead as a whole)
            float dq[EPL], dot = 0.0f;
#pragma unroll
            for (int c = 0; c < EPL; ++c) {
                dq[c] = qi.v[c] - qj.v[c];
                dot = fmaf(p.v[c], dq[c], dot);
            }
            const float x = group_sum(dot);
            float ope;
            const float sneg = sigmoid_neg(x, ope);            // sigmoid(-x)
            const float g = -sneg * inv_batch;                 // dL/dx
            if (loss_acc != nullptr && k == 0) loss_local += softplus_neg(x, ope);
            // item gradients (shared rows): G[i] += g p ; G[j] -= g p
            if constexpr ((PASS & kPassItems) != 0) {
                const float gi = RSX_ABL(256) ? g * 1.01f : g;   // (dev build only: a planted 1 % error, tests/test_mutation.py)
                if (!RSX_ABL(1)) {
                    if (has_hot && hs >= 0)       // popular item: one of its private replica rows (a small table: 32-bit offsets)
                        p.atomic_axpy_at(hot.ghot, row_off<D, uint32_t>(hs * hot.replicas + (int32_t)(wave & (hot.replicas - 1)), k), gi);
                    else
                        p.atomic_axpy_at(G, i_off, gi);
                }
                if (!RSX_ABL(2)) p.atomic_axpy_at(G, j_off, -gi);
                if (hot.touched != nullptr && k == 0) {       // (replicated rows are always visited by the apply: no mark)
                    if (!(has_hot && hs >= 0)) hot.touched[i] = 1;
                    hot.touched[j] = 1;
                }
            }
            // user row: P[u] -= lr * g * (qi - qj)
            const float s = -lr * g * (RSX_ABL(128) ? 1.01f : 1.0f);
            if constexpr (MODE == 0) {
                if constexpr ((PASS & kPassUsers) != 0) {
#pragma unroll
                    for (int c = 0; c < EPL; ++c) p.v[c] = fmaf(s, dq[c], p.v[c]);
                    if (!RSX_ABL(4)) p.store_once_at(P, u_off);
                }
            } else {
                Row<D> dr;
#pragma unroll
                for (int c = 0; c < EPL; ++c) dr.v[c] = dq[c];
                if constexpr (MODE == 1) {
                    const int32_t slot = owner[u] - 1;
                    dr.atomic_axpy(GU + (size_t)slot * D, k, s);
                } else {
                    dr.atomic_axpy(GU + (size_t)u * D, k, g);
                }
            }
        }
        // the next triplet's rows, behind this trip's atomics and stores; then the indices of the one after
        live_n = (bn < B) && (in >= 0);
        request_rows();
        um = im = jm = -1;
        if (bn + stride < B) { um = U_idx[bn + stride]; im = I_idx[bn + stride]; jm = J_idx[bn + stride]; }
        b = bn;
    }
    if (loss_acc != nullptr) {
        const float w = wave_sum(loss_local);
        if (lane == 0) rsx_atomic_add(loss_acc + (wave & 63) * (RSX_LOSS_SLOTS / 64), w);   // one 128-B line per slot
    }
}

// ---- blocked negatives + run-length positives ----------------------------------------------
// When the batch is large against the catalog (B >= 2 I) every item row receives several
// updates per step and the fp32 atomic unit is the bound.  The sampler can
// order the batch so that almost all of those updates are summed on chip first:
//  * rsx_bpr_sample(neg_block = c) draws the negative of batch position p from the item block
//        pi(w),  w = floor(floor(p I / B) / c),   pi = keyed permutation of the blocks
//    (a user's position is uniform, so each user still sees a uniformly distributed negative).
//    Wavefront w of this kernel owns exactly the positions of block w, so ALL negative-side
//    gradients of its c item rows are summed in a wave-private LDS tile (plain read-modify-write)
//    and leave the CU once per row instead of once per triplet.
//  * with RSX_SAMPLE_SORT_POS the positions are sorted by positive item; each lane group walks a
//    CONTIGUOUS range and keeps the running sum of g*P[u] for the current positive item in
//    registers, touching G once per run.
// Neither is a correctness contract: a negative outside the wave's block and an unsorted batch
// take the global-atomic path / runs of length one, and the sums are the same.
#ifndef RSX_BLOCKED_WAVES
#define RSX_BLOCKED_WAVES 6     // wavefronts per SIMD the blocked kernel is compiled for
#endif
#ifndef RSX_PLAIN_BLOCKS_PER_CU
#define RSX_PLAIN_BLOCKS_PER_CU 8   // plain kernel: workgroups per CU the grid is capped at (development A/B)
#endif
#ifndef RSX_RUNS_ROUNDS
#define RSX_RUNS_ROUNDS 2       // TILE = false: rounds of wavefronts the batch is cut into.  Same box, kernel us at 1 / 2 / 4 / 8 rounds:
                                // independent negatives B = 1M 568 / 523 / 532 / 603, configs[3] slice 826 / 771 / 772 / 815 (round 3)
#endif
#ifndef RSX_STEP_PIPELINE
#define RSX_STEP_PIPELINE 1     // 0: the round-2 trip loop (development A/B)
#endif
// TILE = false is the same walk without the negative-side LDS tile, for batches that are ordered by
// positive item but too small for blocked negatives (B < 2 I: fewer than two updates per item row, so the
// negative side has nothing to sum): wavefront w owns the positions [w * span, (w + 1) * span), positive runs
// are summed in registers (Zipf positives at B = 65 536: 65K row updates become ~25K), negatives go to G one by one.
// ITEM CHUNKS (include/rsx.h; chunks.C > 1, TILE only).  nbc wavefronts belong to each item range k: they take the batch
// positions [chunk_pos[k], chunk_pos[k + 1]) -- all triplets whose positive lies in the range, and by the sampler's rule their
// negatives too -- so they read and write item rows [k * Ic, (k + 1) * Ic) ONLY.  A launch covers the ranges [first, first + count):
// the native loop launches every range on a stream of its own, which makes a range's gradients, exchange and apply a pipeline
// independent of the other ranges'.
struct ChunkRun {
    int C;                       // <= 1: off
    int first, count;            // the ranges this launch covers
    int64_t Ic, nbc;             // rows and negative blocks per range
    const int64_t *pos;          // [C + 1] first batch position of each range (device, written by the sampler)
    uint32_t *progress;          // [RSX_PROGRESS_WORDS]: contract violations (RSX_PROGRESS_VIOLATIONS)
};

// (D = 256, round 5: a row is 8 floats per lane -- six rows in flight are 48 registers before anything else -- so the kernel is
//  compiled for 4 wavefronts per SIMD there; the BASELINE shapes, d <= 128, keep RSX_BLOCKED_WAVES.  The 64-bit-offset form --
//  tables of 4 GB and more -- gets one wavefront less than that: at 6 it spilled 6 VGPRs into scratch inside the trip loop)
template <int D, int PASS, typename OffT, bool TILE>
__global__ __launch_bounds__(kBlock, (D > 128 ? 4 : (sizeof(OffT) == 8 ? RSX_BLOCKED_WAVES - 1 : RSX_BLOCKED_WAVES))) void bpr_step_blocked_kernel(
    float *__restrict__ P, const float *__restrict__ Q, float *__restrict__ G,
    const int32_t *__restrict__ U_idx, const int32_t *__restrict__ I_idx,
    const int32_t *__restrict__ J_idx, int64_t B, int64_t num_items, int c, int64_t span, uint64_t neg_key, float lr,
    float inv_batch, float *__restrict__ loss_acc, HotMap hot, ChunkRun chunks)
{
    constexpr bool kItems = (PASS & kPassItems) != 0, kUsers = (PASS & kPassUsers) != 0;
    extern __shared__ __attribute__((aligned(16))) float neg_acc[];   // [4 waves][c][D]
    constexpr int EPL = D / 32;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: everything derived from it lives in SGPRs
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wib;
    float *acc = nullptr;
    int64_t b0, b1;
    int32_t item_lo = 0, item_hi = 0;                // the item block of this wavefront's negatives (TILE)
    int32_t range_lo = 0, range_hi = 0x7fffffff;     // item range every row this wavefront sums into must lie in (chunks)
    int my_range = 0;
    if constexpr (TILE) {
        if (chunks.C > 1) {
            const int local = (int)((uint32_t)wave / (uint32_t)chunks.nbc);
            if (local >= chunks.count) return;
            my_range = chunks.first + local;
            const int64_t wic = wave - (int64_t)local * chunks.nbc;
            const int64_t pc = chunks.pos[my_range], nc = chunks.pos[my_range + 1] - pc;
            b0 = pc + ceil_div64(wic * c * nc, chunks.Ic);
            b1 = pc + ceil_div64((wic + 1) * c * nc, chunks.Ic);
            range_lo = (int32_t)(my_range * chunks.Ic);
            range_hi = (int32_t)(range_lo + chunks.Ic);
            item_lo = range_lo + (int32_t)(neg_block_of(wic, chunks.nbc, chunk_key(neg_key, my_range)) * c);
            item_hi = item_lo + c;                   // (padding rows of the range exist in G and stay zero)
        } else {
        const int64_t nblocks = ceil_div64(num_items, c);
        if (wave >= nblocks) return;
        // batch positions of this wavefront, and the item block its negatives come from
        const int64_t nom_lo = wave * c;
        const int64_t nom_hi = (nom_lo + c < num_items) ? nom_lo + c : num_items;
        b0 = ceil_div64(nom_lo * B, num_items);
        b1 = ceil_div64(nom_hi * B, num_items);
        item_lo = (int32_t)(neg_block_of(wave, nblocks, neg_key) * c);            // num_items < 2^31
        item_hi = (int32_t)(((int64_t)item_lo + c < num_items) ? (int64_t)item_lo + c : num_items);
        }
        acc = neg_acc + (size_t)wib * c * D;
        for (int e = lane; e < c * D; e += 64) acc[e] = 0.0f;
    } else if (chunks.C > 1) {
        // item ranges without blocks (B < 2 I): chunks.nbc wavefronts per range share the range's positions evenly; every item
        // row they touch -- positive runs and negatives, both through G -- lies in the range by the sampler's rule
        const int local = (int)((uint32_t)wave / (uint32_t)chunks.nbc);
        if (local >= chunks.count) return;
        my_range = chunks.first + local;
        const int64_t wic = wave - (int64_t)local * chunks.nbc;
        const int64_t pc = chunks.pos[my_range], nc = chunks.pos[my_range + 1] - pc;
        b0 = pc + ceil_div64(wic * nc, chunks.nbc);
        b1 = pc + ceil_div64((wic + 1) * nc, chunks.nbc);
        if (b0 >= b1) return;
        range_lo = (int32_t)(my_range * chunks.Ic);
        range_hi = (int32_t)(range_lo + chunks.Ic);
    } else {
        b0 = wave * span;
        if (b0 >= B) return;
        b1 = (b0 + span < B) ? b0 + span : B;
    }
    // each lane group walks a contiguous part of [b0, b1)
    const int64_t len = ceil_div64(b1 - b0, TPW);
    const int64_t g_lo = b0 + sub * len;
    const int64_t g_hi = (g_lo + len < b1) ? g_lo + len : b1;

    float loss_local = 0.0f;
    // running sum of g*P[u] for the current positive item (the sampler orders the batch by it)
    int32_t run_item = -1;
    int32_t run_hs = -1;         // the run's replica slot (HotMap), fetched with the position's rows: a load at flush time is a
                                 // wait in the middle of the trip, and on gfx950 that wait drains the trip's atomics and stores too
    float run[EPL];
#pragma unroll
    for (int cc = 0; cc < EPL; ++cc) run[cc] = 0.f;
    // a popular item's run spans hundreds of wavefronts, which all flush into the same row at
    // about the same time (same-line atomics serialise): such rows go to the replicas (HotMap)
#define RSX_RUN_FLUSH(RI, R)                                                      \
    if (kItems && RI >= 0 && !RSX_ABL(1)) {                                       \
        if (chunks.C > 1 && (RI < range_lo || RI >= range_hi) && k == 0)          \
            atomicAdd(chunks.progress + RSX_PROGRESS_VIOLATIONS, 1u);            \
        const int32_t hs = run_hs;                                                \
        if (hs >= 0) {                                                            \
            float *grow = at(hot.ghot, row_off<D, uint32_t>(hs * hot.replicas + (int32_t)(wave & (hot.replicas - 1)), k)); \
            _Pragma("unroll") for (int cc = 0; cc < EPL; ++cc) rsx_atomic_add(grow + 32 * cc, R[cc]); \
        } else {                                                                  \
            float *grow = at(G, row_off<D, OffT>(RI, k));                         \
            _Pragma("unroll") for (int cc = 0; cc < EPL; ++cc) rsx_atomic_add(grow + 32 * cc, R[cc]); \
        }                                                                         \
    }

    // one triplet whose three rows are already in registers
    auto process = [&](bool live, int32_t u, int32_t i, int32_t j, int32_t hs_i, Row<D> &p,
                       const Row<D> &qi, const Row<D> &qj) __attribute__((always_inline)) {
        if (!live) return;
        // x = <p, q_i> - <p, q_j> as one dot product with the difference row (the user update needs it anyway): one butterfly
        float dq[EPL], dot = 0.0f;
#pragma unroll
        for (int cc = 0; cc < EPL; ++cc) {
            dq[cc] = qi.v[cc] - qj.v[cc];
            dot = fmaf(p.v[cc], dq[cc], dot);
        }
        const float x = group_sum(dot);
        float ope;
        const float sneg = sigmoid_neg(x, ope);
        const float g = -sneg * inv_batch;
        if (loss_acc != nullptr && k == 0) loss_local += softplus_neg(x, ope);
        if constexpr (kItems) {
            const float gi = RSX_ABL(256) ? g * 1.01f : g;   // (dev build only: a planted 1 % error, tests/test_mutation.py)
            // positive item: extend its run, or flush the run and start a new one
            if (i != run_item) {
                RSX_RUN_FLUSH(run_item, run)
                run_item = i;
                run_hs = hs_i;
#pragma unroll
                for (int cc = 0; cc < EPL; ++cc) run[cc] = 0.f;
            }
#pragma unroll
            for (int cc = 0; cc < EPL; ++cc) run[cc] = fmaf(gi, p.v[cc], run[cc]);
            // negative item: the wave's own block goes to its LDS tile, anything else to G.  The tile
            // is wave-private and updated by plain read-modify-write (ds_add_f32 measured ~120 clk per
            // wave instruction); the lane groups of one wavefront may hit the same row, so they take
            // turns -- LDS executes a wavefront's instructions in order.
            const bool neg_local = TILE && (j >= item_lo && j < item_hi);
            if (!RSX_ABL(2)) {
                if (!neg_local) {
                    p.atomic_axpy_at(G, row_off<D, OffT>(j, k), -gi);
                    if (chunks.C > 1 && (j < range_lo || j >= range_hi) && k == 0) atomicAdd(chunks.progress + RSX_PROGRESS_VIOLATIONS, 1u);
                }
#pragma unroll
                for (int tt = 0; TILE && tt < TPW; ++tt) {
                    if (sub == tt && neg_local) {
                        float *cell = acc + ((j - item_lo) * LPR + k) * EPL;   // lane k's EPL floats
                        float a[EPL];
                        tile_load<EPL>(cell, a);
#pragma unroll
                        for (int cc = 0; cc < EPL; ++cc) a[cc] = fmaf(-gi, p.v[cc], a[cc]);
                        tile_store<EPL>(cell, a);
                    }
                }
            }
        }
        if constexpr (kUsers) {      // last: p is dead after its update
            const float s = -lr * g * (RSX_ABL(128) ? 1.01f : 1.0f);
#pragma unroll
            for (int cc = 0; cc < EPL; ++cc) p.v[cc] = fmaf(s, dq[cc], p.v[cc]);
            if (!RSX_ABL(4)) p.store_once_at(P, row_off<D, OffT>(u, k));
        }
    };

    // two positions per lane group per trip: all six row gathers are in flight together
    // (the gather is latency-bound: ~2 us per dependent trip under load)
    // The walk is in positions RELATIVE to the lane group's first one (32-bit, like the row offsets):
    // the index arrays are then read as base (SGPRs) + 32-bit byte offset too.
    const int32_t n_pos = (int32_t)((g_hi > g_lo) ? g_hi - g_lo : 0);
    const int32_t n_trip = (int32_t)len;
    const OffT pos0 = (OffT)g_lo * 4;
    auto idx = [&](const int32_t *base, int32_t rel) __attribute__((always_inline)) {
        return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(base) + (pos0 + (OffT)((uint32_t)rel * 4u)));
    };
#if RSX_STEP_PIPELINE
    // SOFTWARE PIPELINE over the trips (round 3).  gfx950 has ONE counter (vmcnt) for loads, stores and atomics, and loads
    // and stores complete out of order with each other: whenever a wavefront waits for ANY load while stores are in flight, the
    // wait is a full drain (s_waitcnt vmcnt(0)).  The loop used to be  [indices of the next trip] [row gathers] [wait] [compute,
    // atomics, stores] [copy the prefetched indices: a drain] : two dependent round trips per trip, the gathers of trip t+1 never in
    // flight together with the stores and atomics of trip t.  Now the row gathers of trip t+1 are ISSUED at the end of trip t,
    // behind its stores and atomics and into the same registers (all row registers are dead by then), and the one wait of a trip --
    // at its first use of the rows -- drains both at once.  n* = indices of the trip whose rows are in flight, m* = of the one after.
    // (Round 3, measured and NOT kept: this loop written over arrays of NP positions with NP a template constant.  At NP = 2 the
    //  same registers and no spill, yet 330 -> 392 us per step at the headline and 242 -> 285 at d = 64 on the same box -- the
    //  compiler schedules the unrolled array form differently; NP = 3 / 4 at d <= 64, where rows are short, were slower still for
    //  the blocked kernel.  The explicit a / b form below is the measured one: profiles/r03_exp_sampler_placement.txt, block H.)
    int32_t ua = -1, ia = -1, ja = -1, ub = -1, ib = -1, jb = -1;
    int32_t una = -1, ina = -1, jna = -1, unb = -1, inb = -1, jnb = -1;
    int32_t uma = -1, ima = -1, jma = -1, umb = -1, imb = -1, jmb = -1;
    if (0 < n_pos) { una = idx(U_idx, 0); ina = idx(I_idx, 0); jna = idx(J_idx, 0); }
    if (1 < n_pos) { unb = idx(U_idx, 1); inb = idx(I_idx, 1); jnb = idx(J_idx, 1); }
    Row<D> pa, qia, qja, pb, qib, qjb;
    int32_t hsa = -1, hsb = -1;                      // replica slots of the positions' positive items (with the rows)
    const bool has_hot = kItems && hot.slot != nullptr;
    bool live_a = (0 < n_pos) && (ina >= 0), live_b = (1 < n_pos) && (inb >= 0);
    if (live_a) { pa.load_once_at(P, row_off<D, OffT>(una, k)); qia.load_at(Q, row_off<D, OffT>(RSX_ABL(32) ? 0 : ina, k)); qja.load_at(Q, row_off<D, OffT>(RSX_ABL(64) ? 1 : jna, k)); if (has_hot) hsa = hot.slot[ina]; }
    if (live_b) { pb.load_once_at(P, row_off<D, OffT>(unb, k)); qib.load_at(Q, row_off<D, OffT>(RSX_ABL(32) ? 0 : inb, k)); qjb.load_at(Q, row_off<D, OffT>(RSX_ABL(64) ? 1 : jnb, k)); if (has_hot) hsb = hot.slot[inb]; }
    if (2 < n_pos) { uma = idx(U_idx, 2); ima = idx(I_idx, 2); jma = idx(J_idx, 2); }
    if (3 < n_pos) { umb = idx(U_idx, 3); imb = idx(I_idx, 3); jmb = idx(J_idx, 3); }
    for (int32_t t = 0; t < n_trip; t += 2) {       // wave-uniform trip count
        // (the first use of the prefetched indices / the rows: the trip's one wait)
        ua = una; ia = ina; ja = jna; ub = unb; ib = inb; jb = jnb;
        una = uma; ina = ima; jna = jma; unb = umb; inb = imb; jnb = jmb;
        const bool la = live_a, lb = live_b;
        process(la, ua, ia, ja, hsa, pa, qia, qja);
        process(lb, ub, ib, jb, hsb, pb, qib, qjb);
        // the next trip's rows, behind this trip's stores and atomics
        live_a = (t + 2 < n_pos) && (ina >= 0);
        live_b = (t + 3 < n_pos) && (inb >= 0);
        if (live_a) { pa.load_once_at(P, row_off<D, OffT>(una, k)); qia.load_at(Q, row_off<D, OffT>(RSX_ABL(32) ? 0 : ina, k)); qja.load_at(Q, row_off<D, OffT>(RSX_ABL(64) ? 1 : jna, k)); if (has_hot) hsa = hot.slot[ina]; }
        if (live_b) { pb.load_once_at(P, row_off<D, OffT>(unb, k)); qib.load_at(Q, row_off<D, OffT>(RSX_ABL(32) ? 0 : inb, k)); qjb.load_at(Q, row_off<D, OffT>(RSX_ABL(64) ? 1 : jnb, k)); if (has_hot) hsb = hot.slot[inb]; }
        // ... and the indices of the trip after it
        uma = ima = jma = umb = imb = jmb = -1;
        if (t + 4 < n_pos) { uma = idx(U_idx, t + 4); ima = idx(I_idx, t + 4); jma = idx(J_idx, t + 4); }
        if (t + 5 < n_pos) { umb = idx(U_idx, t + 5); imb = idx(I_idx, t + 5); jmb = idx(J_idx, t + 5); }
    }
#else
    int32_t ua = -1, ia = -1, ja = -1, ub = -1, ib = -1, jb = -1;
    if (0 < n_pos) { ua = idx(U_idx, 0); ia = idx(I_idx, 0); ja = idx(J_idx, 0); }
    if (1 < n_pos) { ub = idx(U_idx, 1); ib = idx(I_idx, 1); jb = idx(J_idx, 1); }
    for (int32_t t = 0; t < n_trip; t += 2) {       // wave-uniform trip count
        int32_t una = -1, ina = -1, jna = -1, unb = -1, inb = -1, jnb = -1;
        if (t + 2 < n_pos) { una = idx(U_idx, t + 2); ina = idx(I_idx, t + 2); jna = idx(J_idx, t + 2); }
        if (t + 3 < n_pos) { unb = idx(U_idx, t + 3); inb = idx(I_idx, t + 3); jnb = idx(J_idx, t + 3); }
        const bool live_a = (t < n_pos) && (ia >= 0);
        const bool live_b = (t + 1 < n_pos) && (ib >= 0);
        Row<D> pa, qia, qja, pb, qib, qjb;
        if (live_a) { pa.load_once_at(P, row_off<D, OffT>(ua, k)); qia.load_at(Q, row_off<D, OffT>(RSX_ABL(32) ? 0 : ia, k)); qja.load_at(Q, row_off<D, OffT>(RSX_ABL(64) ? 1 : ja, k)); }
        if (live_b) { pb.load_once_at(P, row_off<D, OffT>(ub, k)); qib.load_at(Q, row_off<D, OffT>(RSX_ABL(32) ? 0 : ib, k)); qjb.load_at(Q, row_off<D, OffT>(RSX_ABL(64) ? 1 : jb, k)); }
        int32_t hsa = -1, hsb = -1;
        if (kItems && hot.slot != nullptr) { if (live_a) hsa = hot.slot[ia]; if (live_b) hsb = hot.slot[ib]; }
        process(live_a, ua, ia, ja, hsa, pa, qia, qja);
        process(live_b, ub, ib, jb, hsb, pb, qib, qjb);
        ua = una; ia = ina; ja = jna; ub = unb; ib = inb; jb = jnb;
    }
#endif
    RSX_RUN_FLUSH(run_item, run)     // last run of this lane group
#undef RSX_RUN_FLUSH
    // flush the block's rows: one global atomic row per touched item
    const int rows = (kItems && TILE) ? (int)(item_hi - item_lo) : 0;
    for (int m = sub; m - sub < rows; m += TPW) {
        float v[EPL];
#pragma unroll
        for (int cc = 0; cc < EPL; ++cc) v[cc] = 0.f;
        bool nz = false;
        if (m < rows) {
            tile_load<EPL>(acc + (m * LPR + k) * EPL, v);
#pragma unroll
            for (int cc = 0; cc < EPL; ++cc) nz |= (v[cc] != 0.0f);
        }
        const unsigned long long bal = __ballot(nz);
        const unsigned long long gmask = 0xFFFFFFFFull << (sub * 32);
        if (m < rows && (bal & gmask) != 0ull) {
            float *grow = at(G, row_off<D, OffT>(item_lo + m, k));
#pragma unroll
            for (int cc = 0; cc < EPL; ++cc) rsx_atomic_add(grow + 32 * cc, v[cc]);
        }
    }
    if (loss_acc != nullptr) {
        const float w = wave_sum(loss_local);
        if (lane == 0) rsx_atomic_add(loss_acc + (wave & 63) * (RSX_LOSS_SLOTS / 64), w);   // one 128-B line per slot
    }
    // (a per-range "wavefronts done" counter was kept here at first: one atomic per wavefront onto ONE address serialises at the
    //  memory-side atomic unit -- ~40 same-line operations per microsecond, DESIGN.md 4.1 -- and with 16 667 wavefronts cost the
    //  chunked kernel 25 us, 300 us with blocks of 3 items: removed)
}

// ---- the pointwise branch of the reference model (models/MF.py:99-102 with hparams['pointwise'] = True) -------------
// loss = loss_func(<P[u], Q[i]>, rating), mean over the batch; loss_func = F.binary_cross_entropy_with_logits
// (KIND 0; hparams['loss_func'] != 'mse') or F.mse_loss (KIND 1), MF.py:21.  Dense gradients like rsx_bpr_grad:
// users AND items repeat inside a batch (the reference's generator adds one negative for every user to every batch,
// data/generators.py:119-124), so both sides go through atomics into GP / GQ.
template <int D, int KIND, typename OffT>
__global__ __launch_bounds__(kBlock) void pointwise_grad_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                                float *__restrict__ GP, float *__restrict__ GQ,
                                                                const int32_t *__restrict__ U_idx,
                                                                const int32_t *__restrict__ I_idx,
                                                                const float *__restrict__ Y, int64_t n, float inv_n,
                                                                float *__restrict__ loss_acc)
{
    constexpr int EPL = D / 32;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * TPW;
    float loss_local = 0.0f;
    for (int64_t b = wave * TPW + sub; b - sub < n; b += stride) {      // wave-uniform trip count
        if (b >= n) continue;
        const int32_t ib = I_idx[b];
        if (ib < 0) continue;          // no item (a sampler slot of a user without a usable row): skipped like the BPR kernels do
        const OffT u_off = row_off<D, OffT>(U_idx[b], k), i_off = row_off<D, OffT>(ib, k);
        const float y = Y[b];
        Row<D> p, q;
        p.load_at(P, u_off);
        q.load_at(Q, i_off);
        float x = 0.0f;
#pragma unroll
        for (int c = 0; c < EPL; ++c) x = fmaf(p.v[c], q.v[c], x);
        x = group_sum(x);
        float g;
        if constexpr (KIND == 1) {
            g = 2.0f * (x - y) * inv_n;
            if (loss_acc != nullptr && k == 0) loss_local += (x - y) * (x - y);
        } else {
            float ope;
            g = (sigmoid_neg(-x, ope) - y) * inv_n;                                  // sigmoid(x) - y
            if (loss_acc != nullptr && k == 0) loss_local += softplus_neg(-x, ope) - x * y;   // max(x, 0) - x y + log(1 + exp(-|x|))
        }
        if (RSX_ABL(1024)) g *= 1.01f;  // (dev build only: a planted 1 % error, tests/test_mutation.py)
        if (GP != nullptr) {           // (NULL gradient buffers: the loss alone -- MF.process_one_batch on the pointwise branch)
            q.atomic_axpy_at(GP, u_off, g);
            p.atomic_axpy_at(GQ, i_off, g);
        }
    }
    if (loss_acc != nullptr) {
        const float w = wave_sum(loss_local);
        if (lane == 0) rsx_atomic_add(loss_acc + (wave & 63) * (RSX_LOSS_SLOTS / 64), w);
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
    constexpr int L4 = D / 4;          // lanes per row with one float4 each
    constexpr int G4 = 64 / L4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / L4;
    const int k = lane % L4;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * G4;
    for (int64_t b = wave * G4 + sub; b < B; b += stride) {
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

// dense Adam sweep as torch's single-tensor Adam does it (models/MF.py:30: lr 1e-3, betas
// (0.9, 0.999), eps 1e-8, no weight decay), then G = 0.  Every element moves, also where g == 0.
// The scalars arrive as torch's kernels receive them: 1 - beta formed in double on the host and rounded once.
__global__ __launch_bounds__(kBlock) void adam_apply_kernel(float4 *__restrict__ W, float4 *__restrict__ M,
                                                            float4 *__restrict__ V, float4 *__restrict__ G,
                                                            int64_t n4, float one_minus_beta1, float beta2, float one_minus_beta2,
                                                            float eps, float step_size, float bc2_sqrt)
{
    for (int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x; n < n4; n += (int64_t)gridDim.x * kBlock) {
        const float4 g = G[n];
        float4 m = M[n], v = V[n], w = W[n];
#define RSX_ADAM1(c)                                             \
        m.c = m.c + one_minus_beta1 * (g.c - m.c);               \
        v.c = beta2 * v.c + one_minus_beta2 * g.c * g.c;         \
        w.c = w.c - step_size * (m.c / (sqrtf(v.c) / bc2_sqrt + eps));
        RSX_ADAM1(x) RSX_ADAM1(y) RSX_ADAM1(z) RSX_ADAM1(w)
#undef RSX_ADAM1
        M[n] = m; V[n] = v; W[n] = w;
        if (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f) G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
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
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * TPW;
    for (int64_t b = wave * TPW + sub; b - sub < n; b += stride) {
        float acc = 0.0f;
        if (b < n) {
            Row<D> p, q;
            p.load(P + (size_t)U_idx[b] * D, k);
            q.load(Q + (size_t)I_idx[b] * D, k);
#pragma unroll
            for (int c = 0; c < D / 32; ++c) acc = fmaf(p.v[c], q.v[c], acc);
        }
        acc = group_sum(acc);
        if (b < n && k == 0) out[b] = acc;
    }
}

// Q -= lr*G ; G = 0   (streaming; rows with an all-zero gradient quad are not written).
// With a HotMap the replicas of a popular row are summed in here (and zeroed), which saves the
// separate fold launch when no all-reduce sits between the step and the apply.
// DENSE (the native loop, batches of at least one triplet per item: nearly every row has a gradient): the Q quads are requested
// TOGETHER with the G quads instead of after them -- one memory latency per thread instead of two.
template <bool DENSE>
__global__ __launch_bounds__(kBlock) void apply_item_grad_kernel(float4 *__restrict__ Q,
                                                                 float4 *__restrict__ G, int64_t n4,
                                                                 float lr, HotMap hot, int d4)
{
    // four quads per thread per trip: the four G loads are in flight together, then the Q loads of
    // the quads that have a gradient (at small batches most rows have none and are never read)
    constexpr int kQuads = 4;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t n0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; n0 < n4; n0 += stride * kQuads) {
        float4 g[kQuads], q[kQuads];
#pragma unroll
        for (int c = 0; c < kQuads; ++c) {
            const int64_t n = n0 + c * stride;
            g[c] = make_float4(0.f, 0.f, 0.f, 0.f); q[c] = g[c];
            if (n < n4) { g[c] = G[n]; if constexpr (DENSE) q[c] = Q[n]; }
        }
        // (a replicated row whose G quad and replicas cancel EXACTLY -- a triplet with i == j, which the reference's generator can
        //  draw, data/generators.py:169-190: +g p in the replica, -g p in G -- has nothing to apply, but its G quad must still be
        //  cleared: the replicas are.  Found by tools/fuzz_campaign.sh, round 5)
        bool cancel[kQuads] = {false, false, false, false};
        if (hot.slot != nullptr) {
#pragma unroll
            for (int c = 0; c < kQuads; ++c) {
                const int64_t n = n0 + c * stride;
                if (n >= n4) continue;
                const int64_t row = n / d4;
                const int32_t hs = hot.slot[row];
                if (hs >= 0) {
                    cancel[c] = g[c].x != 0.f || g[c].y != 0.f || g[c].z != 0.f || g[c].w != 0.f;
                    float4 *src = reinterpret_cast<float4 *>(hot.ghot) + ((size_t)hs * hot.replicas) * d4 + (n - row * d4);
                    for (int r = 0; r < hot.replicas; ++r) {
                        const float4 v = src[(size_t)r * d4];
                        g[c].x += v.x; g[c].y += v.y; g[c].z += v.z; g[c].w += v.w;
                        src[(size_t)r * d4] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
        bool nz[kQuads];
#pragma unroll
        for (int c = 0; c < kQuads; ++c) {
            const int64_t n = n0 + c * stride;
            nz[c] = n < n4 && (g[c].x != 0.f || g[c].y != 0.f || g[c].z != 0.f || g[c].w != 0.f);
            if constexpr (!DENSE) { if (nz[c]) q[c] = Q[n]; }
        }
#pragma unroll
        for (int c = 0; c < kQuads; ++c) {
            const int64_t n = n0 + c * stride;
            if (nz[c]) {
                float4 w = q[c];
                w.x = fmaf(-lr, g[c].x, w.x); w.y = fmaf(-lr, g[c].y, w.y);
                w.z = fmaf(-lr, g[c].z, w.z); w.w = fmaf(-lr, g[c].w, w.w);
                Q[n] = w;
                G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (cancel[c]) {
                G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

// G[hot_items[s]] += sum_r ghot[s][r] ; ghot = 0     (one lane group per hot row)
template <int D>
__global__ __launch_bounds__(kBlock) void fold_hot_kernel(float *__restrict__ G, float *__restrict__ ghot,
                                                          const int32_t *__restrict__ hot_items, int n_hot,
                                                          int replicas)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int s = t / (D / 4), k = t % (D / 4);
    if (s >= n_hot) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *src = reinterpret_cast<float4 *>(ghot + (size_t)s * replicas * D) + k;
    for (int r = 0; r < replicas; ++r) {
        const float4 v = src[(size_t)r * (D / 4)];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        src[(size_t)r * (D / 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 *dst = reinterpret_cast<float4 *>(G + (size_t)hot_items[s] * D) + k;
    float4 g = *dst;
    g.x += acc.x; g.y += acc.y; g.z += acc.z; g.w += acc.w;
    *dst = g;
}

// the same for the hot rows inside [row_lo, row_hi) only (one item range of the chunked step)
template <int D>
__global__ __launch_bounds__(kBlock) void fold_hot_range_kernel(float *__restrict__ G, float *__restrict__ ghot,
                                                                const int32_t *__restrict__ hot_items, int n_hot,
                                                                int replicas, int64_t row_lo, int64_t row_hi)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int s = t / (D / 4), k = t % (D / 4);
    if (s >= n_hot) return;
    const int64_t item = hot_items[s];
    if (item < row_lo || item >= row_hi) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *src = reinterpret_cast<float4 *>(ghot + (size_t)s * replicas * D) + k;
    for (int r = 0; r < replicas; ++r) {
        const float4 v = src[(size_t)r * (D / 4)];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        src[(size_t)r * (D / 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 *dst = reinterpret_cast<float4 *>(G + (size_t)item * D) + k;
    float4 g = *dst;
    g.x += acc.x; g.y += acc.y; g.z += acc.z; g.w += acc.w;
    *dst = g;
}

// Q -= lr*G ; G = 0 for the rows the step kernel MARKED (and the replicated rows, whose replicas are summed in here), one float4 per thread,
// D / 4 threads per row; the marks are cleared.  A batch far below the catalog touches few rows: the streaming apply above reads all of G
// to find them (51 MB at 100 000 x 128 -- 22-28 us whether the batch held 4 096 or 65 536 triplets), this one reads a byte per row.
// Same arithmetic per quad as apply_item_grad_kernel: the tables come out bit-identical.
template <int D>
__global__ __launch_bounds__(kBlock) void apply_item_grad_touched_kernel(float4 *__restrict__ Q, float4 *__restrict__ G, int64_t num_items,
                                                                         float lr, HotMap hot)
{
    constexpr int d4 = D / 4;
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t row = t / d4;
    const int k = (int)(t % d4);
    if (row >= num_items) return;
    const int32_t hs = hot.slot != nullptr ? hot.slot[row] : -1;
    if (hs < 0 && hot.touched[row] == 0) return;
    const int64_t n = row * d4 + k;
    float4 g = G[n];
    bool cancel = false;
    if (hs >= 0) {
        cancel = g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f;
        float4 *src = reinterpret_cast<float4 *>(hot.ghot) + ((size_t)hs * hot.replicas) * d4 + k;
        for (int r = 0; r < hot.replicas; ++r) {
            const float4 v = src[(size_t)r * d4];
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
            src[(size_t)r * d4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f) {
        float4 w = Q[n];
        w.x = fmaf(-lr, g.x, w.x); w.y = fmaf(-lr, g.y, w.y);
        w.z = fmaf(-lr, g.z, w.z); w.w = fmaf(-lr, g.w, w.w);
        Q[n] = w;
        G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (cancel) {
        G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (every thread of the row has read the mark above before any of them clears it: the row's threads are lanes of ONE wavefront -- d4 <= 64
    //  divides the wavefront -- and execute in lockstep)
    if (k == 0) hot.touched[row] = 0;          // (a replicated row carries a mark when it was somebody's NEGATIVE)
}

// tables of 4 GB and more need 64-bit row offsets (see ADDRESSING above)
bool wide_offsets(int64_t num_users, int64_t num_items, int d)
{
    const int64_t rows = num_users > num_items ? num_users : num_items;
    return rows * (int64_t)d * 4 >= (1ll << 32);
}

template <int D, int MODE, int PASS, typename OffT>
void launch_step(float *P, const float *Q, float *G, const int32_t *u, const int32_t *i,
                 const int32_t *j, int64_t B, float lr, float inv_batch, float *loss_acc,
                 const int32_t *owner, float *GU, HotMap hot, hipStream_t st)
{
    const int64_t waves = (B + TPW - 1) / TPW;
    int64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = (int64_t)rsx_num_cus() * RSX_PLAIN_BLOCKS_PER_CU;   // 8 blocks x 4 waves = 32 waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((bpr_step_kernel<D, MODE, PASS, OffT>), dim3((unsigned)blocks), dim3(kBlock), 0, st, P, Q,
                       G, u, i, j, B, lr, inv_batch, loss_acc, owner, GU, hot);
}

template <int MODE, int PASS>
void dispatch_step(int d, bool wide, float *P, const float *Q, float *G, const int32_t *u, const int32_t *i,
                   const int32_t *j, int64_t B, float lr, float inv_batch, float *loss_acc,
                   const int32_t *owner, float *GU, HotMap hot, hipStream_t st)
{
#define RSX_LAUNCH(D_) do { if (wide) launch_step<D_, MODE, PASS, uint64_t>(P, Q, G, u, i, j, B, lr, inv_batch, loss_acc, owner, GU, hot, st); \
                            else launch_step<D_, MODE, PASS, uint32_t>(P, Q, G, u, i, j, B, lr, inv_batch, loss_acc, owner, GU, hot, st); } while (0)
    switch (d) {
    case 32: RSX_LAUNCH(32); break;
    case 64: RSX_LAUNCH(64); break;
    case 128: RSX_LAUNCH(128); break;
    default: RSX_LAUNCH(256); break;
    }
#undef RSX_LAUNCH
}

// Resident wavefronts per SIMD of the blocked kernel, by LDS reservation (a 256-thread workgroup = one wavefront per SIMD; the CU's
// LDS -- 160 KB on MI355X -- is read from the device).  Round 4: with the transcendental-unit loss the kernel needs 64 VGPRs at d = 128 (47 at d = 64) and would run 8
// wavefronts per SIMD -- its own duration hardly changes between 5 and 8, but the sampler of the next step, which runs in what the
// step kernel leaves free, is squeezed out and the step's PERIOD grows (same box, headline: 8 resident -> 337-352 us per step
// against 314-317 at 6; profiles/r04_exp_step_valu.txt).  rsx_set_option("step_waves") overrides.
#ifndef RSX_STEP_WAVES_D128
#define RSX_STEP_WAVES_D128 6
#endif
#ifndef RSX_STEP_WAVES_D64
#define RSX_STEP_WAVES_D64 7
#endif
static size_t lds_for_residency(size_t need, int d)
{
    int waves = g_rsx_step_waves > 0 ? g_rsx_step_waves : (d >= 128 ? RSX_STEP_WAVES_D128 : RSX_STEP_WAVES_D64);
    // the CU's LDS as the runtime reports it (MI355X: 160 KB); unknown: no reservation -- the kernel then runs at whatever its
    // registers allow, the result is the same
    const size_t cu = (size_t)rsx_lds_per_cu();
    if (waves >= 8 || cu == 0) return need;
    // the smallest reservation with which waves + 1 workgroups no longer fit the CU (allocation granule 512 B... 1 KB to be safe)
    size_t lds = cu / (size_t)(waves + 1) + 1024;
    lds = (lds + 1023) / 1024 * 1024;
    if (lds * (size_t)waves > cu) lds = cu / (size_t)waves / 1024 * 1024;
    return lds > need ? lds : need;
}

template <int PASS, bool TILE>
void dispatch_blocked(int d, bool wide, unsigned blocks, size_t lds, hipStream_t st, float *P, const float *Q, float *G,
                      const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, int64_t num_items,
                      int c, int64_t span, uint64_t neg_key, float lr, float inv_batch, float *loss_acc, HotMap hot,
                      ChunkRun chunks = ChunkRun{1, 0, 1, 0, 0, nullptr, nullptr})
{
    lds = lds_for_residency(lds, d);
#define RSX_LAUNCH(D_) do { if (wide) hipLaunchKernelGGL((bpr_step_blocked_kernel<D_, PASS, uint64_t, TILE>), dim3(blocks), dim3(kBlock), lds, st, P, Q, G, u, i, j, B, num_items, c, span, neg_key, lr, inv_batch, loss_acc, hot, chunks); \
                            else hipLaunchKernelGGL((bpr_step_blocked_kernel<D_, PASS, uint32_t, TILE>), dim3(blocks), dim3(kBlock), lds, st, P, Q, G, u, i, j, B, num_items, c, span, neg_key, lr, inv_batch, loss_acc, hot, chunks); } while (0)
    switch (d) {
    case 32: RSX_LAUNCH(32); break;
    case 64: RSX_LAUNCH(64); break;
    case 128: RSX_LAUNCH(128); break;
    default: RSX_LAUNCH(256); break;
    }
#undef RSX_LAUNCH
}

int64_t grid_1d(int64_t n)
{
    int64_t blocks = (n + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)rsx_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

}  // namespace

#ifdef RSX_ABLATE
// dev build only (librsx_dev.so): write/load switches for tools/ablate*.py and the planted errors of tests/test_mutation.py
//   128 / 256   1 % on the user-row update / on the item gradients of the step kernels
//   512         Adam: the bias corrections of step t + 1 instead of t (host side)
//   1024        the pointwise branch: 1 % on dL/dx
static int g_ablate_host = 0;
RSX_API int rsx_debug_set_ablation(int mask)
{
    g_ablate_host = mask;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_rsx_ablate), &mask, sizeof(int)) == hipSuccess ? RSX_OK : RSX_E_HIP;
}
#define RSX_ABL_HOST(mask) ((g_ablate_host & (mask)) != 0)
#else
#define RSX_ABL_HOST(mask) false
#endif

RSX_API int64_t rsx_bpr_step_workspace(int64_t num_users, int64_t max_batch, int d)
{
    if (num_users < 0 || max_batch < 0 || !rsx_dim_ok(d)) return RSX_E_INVALID;
    const int64_t owner_bytes = ((num_users * 4 + 255) / 256) * 256;
    return owner_bytes + max_batch * d * 4;
}

RSX_API int rsx_bpr_step(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items,
                         const int32_t *u_dev, const int32_t *i_dev, const int32_t *j_dev,
                         int64_t batch, int d, float lr, float inv_batch, float *loss_acc,
                         unsigned flags, void *ws, int64_t ws_bytes, const int32_t *hot_slot_dev,
                         float *G_hot, int hot_replicas, int neg_block, uint64_t neg_key, rsx_stream_t stream)
{
    return rsx_bpr_step_ex(P, Q, G, num_users, num_items, u_dev, i_dev, j_dev, batch, d, lr, inv_batch, loss_acc, flags, ws, ws_bytes,
                           hot_slot_dev, G_hot, hot_replicas, neg_block, neg_key, nullptr, stream);
}

// (internal, rsx_common.h) the same with the native loop's row marks: `touched_dev` (nullable, [num_items] bytes, zero) is written by the
// plain in-place kernel only -- RSX_USERS_UNIQUE without neg_block / RSX_BATCH_SORTED / a split pass -- and must be NULL otherwise
int rsx_bpr_step_ex(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items,
                    const int32_t *u_dev, const int32_t *i_dev, const int32_t *j_dev,
                    int64_t batch, int d, float lr, float inv_batch, float *loss_acc,
                    unsigned flags, void *ws, int64_t ws_bytes, const int32_t *hot_slot_dev,
                    float *G_hot, int hot_replicas, int neg_block, uint64_t neg_key, uint8_t *touched_dev, rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && (G || (flags & RSX_NO_UPDATE)), "null table pointer");
    RSX_CHECK_ARG(touched_dev == nullptr || ((flags & RSX_USERS_UNIQUE) && neg_block == 0 &&
                                             !(flags & (RSX_NO_UPDATE | RSX_DETERMINISTIC | RSX_ITEMS_ONLY | RSX_USERS_ONLY | RSX_BATCH_SORTED))),
                  "row marks are written by the plain in-place kernel only");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(batch >= 0 && num_users > 0 && num_items > 0, "negative size");
    if (batch == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev && j_dev, "null index pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool wide = wide_offsets(num_users, num_items, d) || batch >= (1ll << 30) || (flags & RSX_WIDE_OFFSETS) != 0;
    HotMap hot{nullptr, nullptr, 1};
    if (hot_slot_dev != nullptr) {
        RSX_CHECK_ARG(G_hot != nullptr, "hot_slot_dev given without G_hot");
        RSX_CHECK_ARG(hot_replicas >= 1 && (hot_replicas & (hot_replicas - 1)) == 0, "hot_replicas must be a power of two");
        hot = HotMap{hot_slot_dev, G_hot, hot_replicas};
    }
    hot.touched = touched_dev;
    if (flags & RSX_NO_UPDATE) {   // loss only: the in-place kernel with neither side written
        dispatch_step<0, 0>(d, wide, P, Q, G, u_dev, i_dev, j_dev, batch, lr, inv_batch, loss_acc, nullptr, nullptr,
                            HotMap{nullptr, nullptr, 1}, st);
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    RSX_CHECK_ARG(neg_block >= 0 && neg_block <= kMaxNegBlock, "neg_block must be in [0, 16]");
    if (flags & RSX_DETERMINISTIC) {
        RSX_CHECK_ARG(flags & RSX_USERS_UNIQUE, "RSX_DETERMINISTIC needs RSX_USERS_UNIQUE");
        RSX_CHECK_ARG(!(flags & (RSX_ITEMS_ONLY | RSX_USERS_ONLY)), "RSX_DETERMINISTIC runs the whole step");
        return rsx_bpr_step_deterministic(P, Q, G, num_items, u_dev, i_dev, j_dev, batch, d, lr, inv_batch, loss_acc, ws,
                                          ws_bytes, st);
    }
    int pass = kPassBoth;          // two-pass step: which side this launch writes
    if (flags & (RSX_ITEMS_ONLY | RSX_USERS_ONLY)) {
        RSX_CHECK_ARG(flags & RSX_USERS_UNIQUE, "RSX_ITEMS_ONLY / RSX_USERS_ONLY need RSX_USERS_UNIQUE");
        RSX_CHECK_ARG((flags & (RSX_ITEMS_ONLY | RSX_USERS_ONLY)) != (RSX_ITEMS_ONLY | RSX_USERS_ONLY), "pick one pass");
        if (flags & RSX_ITEMS_ONLY) pass = kPassItems;
        if (flags & RSX_USERS_ONLY) { pass = kPassUsers; loss_acc = nullptr; }
    }
    if ((flags & RSX_USERS_UNIQUE) && neg_block > 0) {
        const int64_t waves = ceil_div64(num_items, neg_block);
        const unsigned blocks = (unsigned)ceil_div64(waves, kWavesPerBlock);
        const size_t lds = (size_t)kWavesPerBlock * neg_block * d * sizeof(float);
#define RSX_BLOCKED(PASS_) dispatch_blocked<PASS_, true>(d, wide, blocks, lds, st, P, Q, G, u_dev, i_dev, j_dev, batch, num_items, neg_block, 0, neg_key, lr, inv_batch, loss_acc, hot)
        if (pass == kPassItems) RSX_BLOCKED(kPassItems); else if (pass == kPassUsers) RSX_BLOCKED(kPassUsers); else RSX_BLOCKED(kPassBoth);
#undef RSX_BLOCKED
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    if ((flags & RSX_USERS_UNIQUE) && (flags & RSX_BATCH_SORTED)) {
        // RSX_RUNS_ROUNDS rounds of wavefronts (a round = as many as fit the chip at once, 6 per SIMD), each with an even
        // share of the positions, at least 8 (two trips of two positions per lane group)
        const int64_t slots = (int64_t)rsx_num_cus() * 4 * RSX_BLOCKED_WAVES * RSX_RUNS_ROUNDS;
        int64_t span = 2 * ceil_div64(batch, 2 * slots);
        if (span < 8) span = 8;
        const unsigned blocks = (unsigned)ceil_div64(ceil_div64(batch, span), kWavesPerBlock);
#define RSX_RUNS(PASS_) dispatch_blocked<PASS_, false>(d, wide, blocks, 0, st, P, Q, G, u_dev, i_dev, j_dev, batch, num_items, 0, span, 0, lr, inv_batch, loss_acc, hot)
        if (pass == kPassItems) RSX_RUNS(kPassItems); else if (pass == kPassUsers) RSX_RUNS(kPassUsers); else RSX_RUNS(kPassBoth);
#undef RSX_RUNS
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    if (flags & RSX_USERS_UNIQUE) {
#define RSX_INPLACE(PASS_) dispatch_step<0, PASS_>(d, wide, P, Q, G, u_dev, i_dev, j_dev, batch, lr, inv_batch, loss_acc, nullptr, nullptr, hot, st)
        if (pass == kPassItems) RSX_INPLACE(kPassItems); else if (pass == kPassUsers) RSX_INPLACE(kPassUsers); else RSX_INPLACE(kPassBoth);
#undef RSX_INPLACE
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
    dispatch_step<1, kPassBoth>(d, wide, P, Q, G, u_dev, i_dev, j_dev, batch, lr, inv_batch, loss_acc, owner, GU, hot, st);
    const int64_t tpw = 64 / (d / 4);     // bpr_apply_user_kernel: one float4 per lane
    const unsigned g2 = (unsigned)grid_1d((batch + tpw - 1) / tpw * 64);
    switch (d) {
    case 32: hipLaunchKernelGGL(bpr_apply_user_kernel<32>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    case 64: hipLaunchKernelGGL(bpr_apply_user_kernel<64>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    case 128: hipLaunchKernelGGL(bpr_apply_user_kernel<128>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    default: hipLaunchKernelGGL(bpr_apply_user_kernel<256>, dim3(g2), dim3(kBlock), 0, st, P, u_dev, i_dev, batch, owner, GU); break;
    }
    hipLaunchKernelGGL(bpr_release_kernel, dim3(g1), dim3(kBlock), 0, st, u_dev, i_dev, batch, owner);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_bpr_grad(const float *P, const float *Q, float *GP, float *GQ, int64_t num_users,
                         int64_t num_items, const int32_t *u_dev, const int32_t *i_dev,
                         const int32_t *j_dev, int64_t batch, int d, float inv_batch, float *loss_acc,
                         rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && GP && GQ, "null table pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(batch >= 0 && num_users > 0 && num_items > 0, "negative size");
    if (batch == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev && j_dev, "null index pointer");
    dispatch_step<2, kPassBoth>(d, wide_offsets(num_users, num_items, d), const_cast<float *>(P), Q, GQ, u_dev, i_dev, j_dev, batch, 0.0f, inv_batch,
                                loss_acc, nullptr, GP, HotMap{nullptr, nullptr, 1}, (hipStream_t)stream);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_pointwise_grad(const float *P, const float *Q, float *GP, float *GQ, int64_t num_users, int64_t num_items,
                               const int32_t *u_dev, const int32_t *i_dev, const float *y_dev, int64_t n, int d,
                               float inv_n, int loss_kind, float *loss_acc, rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && ((GP && GQ) || (!GP && !GQ && loss_acc)), "null table pointer (GP and GQ may both be NULL for the loss alone)");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(n >= 0 && num_users > 0 && num_items > 0, "negative size");
    RSX_CHECK_ARG(loss_kind == 0 || loss_kind == 1, "loss_kind: 0 = binary cross entropy with logits, 1 = mean squared error");
    if (n == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev && y_dev, "null batch pointer");
    const bool wide = wide_offsets(num_users, num_items, d);
    const unsigned g = (unsigned)grid_1d(ceil_div64(n, TPW) * 64);
    hipStream_t st = (hipStream_t)stream;
#define RSX_PW(D_, K_) do { if (wide) hipLaunchKernelGGL((pointwise_grad_kernel<D_, K_, uint64_t>), dim3(g), dim3(kBlock), 0, st, P, Q, GP, GQ, u_dev, i_dev, y_dev, n, inv_n, loss_acc); \
                            else hipLaunchKernelGGL((pointwise_grad_kernel<D_, K_, uint32_t>), dim3(g), dim3(kBlock), 0, st, P, Q, GP, GQ, u_dev, i_dev, y_dev, n, inv_n, loss_acc); } while (0)
    switch (d * 2 + loss_kind) {
    case 64: RSX_PW(32, 0); break;
    case 65: RSX_PW(32, 1); break;
    case 128: RSX_PW(64, 0); break;
    case 129: RSX_PW(64, 1); break;
    case 256: RSX_PW(128, 0); break;
    case 257: RSX_PW(128, 1); break;
    case 512: RSX_PW(256, 0); break;
    default: RSX_PW(256, 1); break;
    }
#undef RSX_PW
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_adam_apply(float *W, float *M, float *V, float *G, int64_t n, double lr, double beta1,
                           double beta2, double eps, int64_t t, rsx_stream_t stream)
{
    RSX_CHECK_ARG(W && M && V && G, "null pointer");
    RSX_CHECK_ARG(n >= 0 && n % 4 == 0 && t >= 1, "n must be a multiple of 4 and t >= 1");
    RSX_CHECK_ARG(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "bad Adam hyper-parameters");
    if (n == 0) return RSX_OK;
    const double tb = (double)t + (RSX_ABL_HOST(512) ? 1.0 : 0.0);      // (dev build only: a planted error, tests/test_mutation.py)
    const double bc1 = 1.0 - pow(beta1, tb);
    const double bc2 = 1.0 - pow(beta2, tb);
    hipLaunchKernelGGL(adam_apply_kernel, dim3((unsigned)grid_1d(n / 4)), dim3(kBlock), 0, (hipStream_t)stream,
                       (float4 *)W, (float4 *)M, (float4 *)V, (float4 *)G, n / 4, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)eps, (float)(lr / bc1), (float)sqrt(bc2));
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
    const unsigned g = (unsigned)grid_1d((n + TPW - 1) / TPW * 64);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
    case 32: hipLaunchKernelGGL(pair_score_kernel<32>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    case 64: hipLaunchKernelGGL(pair_score_kernel<64>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    case 128: hipLaunchKernelGGL(pair_score_kernel<128>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    default: hipLaunchKernelGGL(pair_score_kernel<256>, dim3(g), dim3(kBlock), 0, st, P, Q, u_dev, i_dev, n, r_out); break;
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_fold_hot_grad(float *G, float *G_hot, const int32_t *hot_items_dev, int n_hot,
                              int hot_replicas, int d, rsx_stream_t stream)
{
    RSX_CHECK_ARG(G && G_hot && hot_items_dev, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && n_hot >= 0 && hot_replicas >= 1, "bad shape");
    if (n_hot == 0) return RSX_OK;
    const int threads = n_hot * (d / 4);
    const unsigned g = (unsigned)((threads + kBlock - 1) / kBlock);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
    case 32: hipLaunchKernelGGL(fold_hot_kernel<32>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas); break;
    case 64: hipLaunchKernelGGL(fold_hot_kernel<64>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas); break;
    case 128: hipLaunchKernelGGL(fold_hot_kernel<128>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas); break;
    default: hipLaunchKernelGGL(fold_hot_kernel<256>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas); break;
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_apply_item_grad(float *Q, float *G, int64_t num_items, int d, float lr,
                                const int32_t *hot_slot_dev, float *G_hot, int hot_replicas,
                                rsx_stream_t stream)
{
    return rsx_apply_item_grad_ex(Q, G, num_items, d, lr, hot_slot_dev, G_hot, hot_replicas, false, (hipStream_t)stream);
}

int rsx_apply_item_grad_ex(float *Q, float *G, int64_t num_items, int d, float lr, const int32_t *hot_slot_dev, float *G_hot,
                           int hot_replicas, bool dense, hipStream_t stream)
{
    RSX_CHECK_ARG(Q && G, "null table pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_items > 0, "bad shape");
    HotMap hot{nullptr, nullptr, 1};
    if (hot_slot_dev != nullptr) {
        RSX_CHECK_ARG(G_hot != nullptr && hot_replicas >= 1, "hot_slot_dev given without G_hot");
        hot = HotMap{hot_slot_dev, G_hot, hot_replicas};
    }
    const int64_t n4 = num_items * d / 4;
    // one trip of four quads per thread when the table is large (an uneven grid-stride tail costs ~25 %)
    const int64_t blocks = ceil_div64(n4, (int64_t)kBlock * 4);
    if (dense) hipLaunchKernelGGL(apply_item_grad_kernel<true>, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(kBlock), 0, stream,
                                  (float4 *)Q, (float4 *)G, n4, lr, hot, d / 4);
    else hipLaunchKernelGGL(apply_item_grad_kernel<false>, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(kBlock), 0, stream,
                            (float4 *)Q, (float4 *)G, n4, lr, hot, d / 4);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

// (internal, rsx_common.h) the apply behind a step that marked its rows (rsx_bpr_step_ex with touched_dev): marked and replicated rows only
int rsx_apply_item_grad_touched(float *Q, float *G, int64_t num_items, int d, float lr, const int32_t *hot_slot_dev, float *G_hot,
                                int hot_replicas, uint8_t *touched_dev, hipStream_t stream)
{
    RSX_CHECK_ARG(Q && G && touched_dev, "null pointer");
    RSX_CHECK_ARG(rsx_dim_ok(d) && num_items > 0, "bad shape");
    HotMap hot{nullptr, nullptr, 1};
    if (hot_slot_dev != nullptr) {
        RSX_CHECK_ARG(G_hot != nullptr && hot_replicas >= 1, "hot_slot_dev given without G_hot");
        hot = HotMap{hot_slot_dev, G_hot, hot_replicas};
    }
    hot.touched = touched_dev;
    const unsigned blocks = (unsigned)ceil_div64(num_items * (d / 4), kBlock);
    switch (d) {
    case 32: hipLaunchKernelGGL(apply_item_grad_touched_kernel<32>, dim3(blocks), dim3(kBlock), 0, stream, (float4 *)Q, (float4 *)G, num_items, lr, hot); break;
    case 64: hipLaunchKernelGGL(apply_item_grad_touched_kernel<64>, dim3(blocks), dim3(kBlock), 0, stream, (float4 *)Q, (float4 *)G, num_items, lr, hot); break;
    case 128: hipLaunchKernelGGL(apply_item_grad_touched_kernel<128>, dim3(blocks), dim3(kBlock), 0, stream, (float4 *)Q, (float4 *)G, num_items, lr, hot); break;
    default: hipLaunchKernelGGL(apply_item_grad_touched_kernel<256>, dim3(blocks), dim3(kBlock), 0, stream, (float4 *)Q, (float4 *)G, num_items, lr, hot); break;
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_bpr_step_chunked(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items,
                                 int64_t items_real, int chunks, const int32_t *u_dev, const int32_t *i_dev,
                                 const int32_t *j_dev, int64_t batch, int d, float lr, float inv_batch, float *loss_acc,
                                 const int32_t *hot_slot_dev, float *G_hot, int hot_replicas, int neg_block,
                                 uint64_t neg_key, const int64_t *chunk_pos_dev, uint32_t *progress_dev,
                                 int first_range, int num_ranges, rsx_stream_t stream)
{
    RSX_CHECK_ARG(P && Q && G && chunk_pos_dev && progress_dev, "null pointer");
    RSX_CHECK_ARG(first_range >= 0 && num_ranges >= 1 && first_range + num_ranges <= chunks, "ranges [first, first + count) must lie in [0, chunks)");
    RSX_CHECK_ARG(rsx_dim_ok(d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(chunks >= 2 && chunks <= RSX_MAX_CHUNKS, "chunks must be in [2, RSX_MAX_CHUNKS]");
    RSX_CHECK_ARG(neg_block >= 0 && neg_block <= kMaxNegBlock, "neg_block must be in [0, 16]");
    RSX_CHECK_ARG(batch >= 0 && num_users > 0 && items_real > 0, "negative size");
    const ChunkGeom g = chunk_geom(items_real, chunks, neg_block < 1 ? 1 : neg_block);
    RSX_CHECK_ARG(num_items == g.Ic * chunks, "num_items must be chunks * rsx_chunk_rows(items_real, chunks, neg_block)");
    if (batch == 0) return RSX_OK;
    RSX_CHECK_ARG(u_dev && i_dev && j_dev, "null index pointer");
    HotMap hot{nullptr, nullptr, 1};
    if (hot_slot_dev != nullptr) {
        RSX_CHECK_ARG(G_hot != nullptr, "hot_slot_dev given without G_hot");
        RSX_CHECK_ARG(hot_replicas >= 1 && (hot_replicas & (hot_replicas - 1)) == 0, "hot_replicas must be a power of two");
        hot = HotMap{hot_slot_dev, G_hot, hot_replicas};
    }
    const bool wide = wide_offsets(num_users, num_items, d) || batch >= (1ll << 30);
    if (neg_block == 0) {
        // no blocks (batches below two triplets per item): the walk of the ordered batch without the negative-side tile, a
        // range's share of RSX_RUNS_ROUNDS rounds of wavefronts per range (at least 8 positions per wavefront of an even split)
        const int64_t slots = (int64_t)rsx_num_cus() * 4 * RSX_BLOCKED_WAVES * RSX_RUNS_ROUNDS;
        int64_t per_range = ceil_div64(slots, chunks);
        const int64_t most = ceil_div64(ceil_div64(batch, chunks), 8);
        if (per_range > most) per_range = most < 1 ? 1 : most;
        const unsigned blocks = (unsigned)ceil_div64(per_range * num_ranges, kWavesPerBlock);
        dispatch_blocked<kPassBoth, false>(d, wide, blocks, 0, (hipStream_t)stream, P, Q, G, u_dev, i_dev, j_dev, batch, num_items,
                                           0, 0, 0, lr, inv_batch, loss_acc, hot,
                                           ChunkRun{chunks, first_range, num_ranges, g.Ic, per_range, chunk_pos_dev, progress_dev});
        RSX_CHECK_LAUNCH();
        return RSX_OK;
    }
    const int64_t waves = g.nbc * num_ranges;
    const unsigned blocks = (unsigned)ceil_div64(waves, kWavesPerBlock);
    const size_t lds = (size_t)kWavesPerBlock * neg_block * d * sizeof(float);
    dispatch_blocked<kPassBoth, true>(d, wide, blocks, lds, (hipStream_t)stream, P, Q, G, u_dev, i_dev, j_dev, batch, num_items,
                                      neg_block, 0, neg_key, lr, inv_batch, loss_acc, hot,
                                      ChunkRun{chunks, first_range, num_ranges, g.Ic, g.nbc, chunk_pos_dev, progress_dev});
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

int rsx_fold_hot_grad_range(float *G, float *G_hot, const int32_t *hot_items_dev, int n_hot, int hot_replicas, int d,
                            int64_t row_lo, int64_t row_hi, hipStream_t st)
{
    if (n_hot == 0) return RSX_OK;
    const int threads = n_hot * (d / 4);
    const unsigned g = (unsigned)((threads + kBlock - 1) / kBlock);
    switch (d) {
    case 32: hipLaunchKernelGGL(fold_hot_range_kernel<32>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas, row_lo, row_hi); break;
    case 64: hipLaunchKernelGGL(fold_hot_range_kernel<64>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas, row_lo, row_hi); break;
    case 128: hipLaunchKernelGGL(fold_hot_range_kernel<128>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas, row_lo, row_hi); break;
    default: hipLaunchKernelGGL(fold_hot_range_kernel<256>, dim3(g), dim3(kBlock), 0, st, G, G_hot, hot_items_dev, n_hot, hot_replicas, row_lo, row_hi); break;
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}
