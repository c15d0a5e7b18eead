// rsx_common.h -- shared helpers of librsx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rsx.h"

#define RSX_API extern "C" __attribute__((visibility("default")))

void rsx_set_error(const char *fmt, ...);

#define RSX_CHECK_ARG(cond, msg)                                      \
    do {                                                              \
        if (!(cond)) {                                                \
            rsx_set_error("%s: invalid argument: %s", __func__, msg); \
            return RSX_E_INVALID;                                     \
        }                                                             \
    } while (0)

#define RSX_CHECK_LAUNCH()                                                           \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            rsx_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return RSX_E_HIP;                                                        \
        }                                                                            \
    } while (0)

static inline bool rsx_dim_ok(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }

// number of CUs on the current device (cached)
int rsx_num_cus();
// LDS bytes per CU of the current device (cached; 0 = unknown)
int rsx_lds_per_cu();

// fp32 hardware atomic add without return (global_atomic_add_f32); the file is
// built with -munsafe-fp-atomics so this never lowers to a CAS loop.
__device__ __forceinline__ void rsx_atomic_add(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Development ablation switches exist ONLY in the dev build (-DRSX_ABLATE -> librsx_dev.so, built by
// `python -m recsys_pytorch_amd.build --dev` for tools/ablate*.py).  In the shipped library RSX_ABL()
// is the constant false: no kernel of librsx.so can be told at run time to drop a store or an atomic.
#ifdef RSX_ABLATE
static __constant__ int c_rsx_ablate = 0;
#define RSX_ABL(mask) ((c_rsx_ablate & (mask)) != 0)
#else
#define RSX_ABL(mask) false
#endif

// process-wide options behind rsx_set_option (include/rsx.h)
extern int g_rsx_score_lanes;   // passes of the fused scoring path in flight (1..4)
extern int g_rsx_sort_cap;      // LDS sort capacity of the bucket sampler (test hook for the out-of-LDS path)
extern int g_rsx_step_waves;    // resident wavefronts per SIMD the blocked step kernel is held to (0: the default of rsx_bpr.hip)
extern int g_rsx_mesh_blocks;   // workgroups of the mesh's two exchange kernels (0: one per CU)
extern int g_rsx_apply_stream;  // chunked + sharded trainer: the ranges' applies on a stream of their own (opt-in)

// rsx_det.hip: the deterministic form of the step (RSX_DETERMINISTIC); arguments validated by rsx_bpr_step
int rsx_bpr_step_deterministic(float *P, const float *Q, float *G, int64_t num_items, const int32_t *u_dev,
                               const int32_t *i_dev, const int32_t *j_dev, int64_t batch, int d, float lr, float inv_batch,
                               float *loss_acc, void *ws, int64_t ws_bytes, hipStream_t st);

// ---- shared by the sampler (rsx_sample.hip) and the step kernels (rsx_bpr.hip) --------------
constexpr int kMaxNegBlock = 16;

__host__ __device__ __forceinline__ int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
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

__host__ __device__ __forceinline__ int half_bits_for(int64_t n)
{
    int bits = 1;
    while ((1ll << bits) < n) ++bits;
    return (bits + 1) / 2;
}

// keyed bijection of [0,n): 4-round Feistel over 2*hb bits with cycle walking
__host__ __device__ __forceinline__ uint32_t feistel_perm(uint32_t x, uint32_t n, int hb, uint64_t key)
{
    const uint32_t mask = (1u << hb) - 1u;
    do {
        uint32_t l = x >> hb, r = x & mask;
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = (uint32_t)splitmix64(key ^ ((uint64_t)r << 8) ^ (uint64_t)round) & mask;
            const uint32_t t = l ^ f;
            l = r; r = t;
        }
        x = (l << hb) | r;
    } while (x >= n);
    return x;
}

// item block whose negatives batch-position block `w` draws from: identity, or a per-step
// keyed permutation of the blocks (needed when the batch is ordered by positive item)
__host__ __device__ __forceinline__ int64_t neg_block_of(int64_t w, int64_t nblocks, uint64_t neg_key)
{
    return neg_key == 0 ? w : (int64_t)feistel_perm((uint32_t)w, (uint32_t)nblocks, half_bits_for(nblocks), neg_key);
}

// ---- item chunks (include/rsx.h: "item chunks") -------------------------------------------------
// The relabelled item space holds C ranges of Ic rows (Ic a multiple of the negative block c); range k has
// real(k) = base + (k < rem) real items at its start, padding rows behind them.  nbc = Ic / c blocks per range.
struct ChunkGeom {
    int C, c;
    int64_t Ic, base, rem, nbc;
    __host__ __device__ __forceinline__ int64_t real(int k) const { return base + (k < rem ? 1 : 0); }
};
__host__ __device__ __forceinline__ ChunkGeom chunk_geom(int64_t items_real, int chunks, int neg_block)
{
    ChunkGeom g;
    g.C = chunks; g.c = neg_block;
    g.base = items_real / chunks; g.rem = items_real % chunks;
    const int64_t most = g.base + (g.rem > 0 ? 1 : 0);
    g.Ic = ceil_div64(most, neg_block) * neg_block;
    g.nbc = g.Ic / neg_block;
    return g;
}
// per-range key of the negative-block permutation (nonzero)
__host__ __device__ __forceinline__ uint64_t chunk_key(uint64_t neg_key, int k) { return splitmix64(neg_key ^ (0xA24BAED4963EE407ull * (uint64_t)(k + 1))) | 1ull; }

// rsx_comm.hip: the collectives the native loop issues (in place on device buffers, asynchronous on `st`)
int rsx_comm_all_reduce(rsx_comm *c, float *buf, int64_t n, hipStream_t st);
int rsx_comm_reduce_scatter(rsx_comm *c, float *buf, int64_t n_per_rank, hipStream_t st);
int rsx_comm_all_gather(rsx_comm *c, float *buf, int64_t n_per_rank, hipStream_t st);

// rsx_sample.hip: was this CSC built from exactly this CSR?
bool rsx_csc_matches(const rsx_csc *c, const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users, int64_t num_items);

// rsx_mesh.hip: the tables a mesh was built over
void rsx_mesh_tables(const rsx_mesh *m, const float **Q, const float **G, int64_t *rows, int *d);

// rsx_bpr.hip: rsx_apply_item_grad with a hint (dense: nearly every row has a gradient; the result is the same either way)
int rsx_apply_item_grad_ex(float *Q, float *G, int64_t num_items, int d, float lr, const int32_t *hot_slot_dev, float *G_hot,
                           int hot_replicas, bool dense, hipStream_t stream);

// rsx_bpr.hip: the step and its apply with the native loop's row marks (small batches: the apply visits the marked rows only)
int rsx_bpr_step_ex(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items, const int32_t *u_dev, const int32_t *i_dev,
                    const int32_t *j_dev, int64_t batch, int d, float lr, float inv_batch, float *loss_acc, unsigned flags, void *ws,
                    int64_t ws_bytes, const int32_t *hot_slot_dev, float *G_hot, int hot_replicas, int neg_block, uint64_t neg_key,
                    uint8_t *touched_dev, rsx_stream_t stream);
int rsx_apply_item_grad_touched(float *Q, float *G, int64_t num_items, int d, float lr, const int32_t *hot_slot_dev, float *G_hot,
                                int hot_replicas, uint8_t *touched_dev, hipStream_t stream);
extern int g_rsx_touched_apply;  // 0 never, 1 (default) where the batch is small against the catalog, 2 wherever the plain kernel runs

// rsx_bpr.hip: pieces of the chunked step the native loop queues on its own stream
int rsx_fold_hot_grad_range(float *G, float *G_hot, const int32_t *hot_items_dev, int n_hot, int hot_replicas, int d,
                            int64_t row_lo, int64_t row_hi, hipStream_t st);
