// rsx_train.hip -- the native batch loop of one epoch of BPR-MF training on one GPU.
//
// Replaces the reference's inner loop (models/MF.py:61-72)
//     for b, (batch_users, batch_pos, batch_neg) in enumerate(batch_generator):
//         self.optimizer.zero_grad(); loss = self.process_one_batch(...); loss.backward(); self.optimizer.step()
//         epoch_loss += batch_loss
// together with the generator that feeds it (data/generators.py:206-224).  One call queues n
// steps on the caller's stream without returning to the interpreter in between:
//     sampler of step t+1 (side stream)  ||  step kernel of step t  ->  [exchange of G]  ->  apply
// Every kernel is the one behind the stand-alone entry points (rsx_bpr_sample, rsx_bpr_step,
// rsx_fold_hot_grad, rsx_apply_item_grad): a run of n steps equals n hand-driven steps.
// The trainer owns only host state (step counter, position in the user permutation, a side
// stream, events); every device buffer is borrowed from the caller.
#include <vector>

#include "rsx_common.h"

// The last item range runs on the caller's stream instead of a stream of its own: the run stream only joins in a chunked run, and
// every HIP stream more is one more client of the runtime's four hardware queues.  Same box, us per step at C = 2 / 3 / 4 ranges:
// a stream per range 350 / 432 / 525, the last range on the run stream 350 / 425 / 488 (profiles/r03_exp_sampler_placement.txt, block Q).
// RSX_SAMPLER_BEHIND_KERNEL = 1 (development A/B, round 5): the sampler of step t + 2 is held back until the step kernel of step t has
// ended -- it then starts beside the apply sweep and the gap behind it, where the chip is emptier, instead of beside the kernel
#ifndef RSX_SAMPLER_BEHIND_KERNEL
#define RSX_SAMPLER_BEHIND_KERNEL 0
#endif
#ifndef RSX_RANGE_ON_RUN_STREAM
#define RSX_RANGE_ON_RUN_STREAM 1
#endif
#if RSX_RANGE_ON_RUN_STREAM
#define RSX_RANGE_STREAM(k) (((k) == c.chunks - 1) ? st : t->cs[k])
#else
#define RSX_RANGE_STREAM(k) (t->cs[k])
#endif

#ifdef RSX_ABLATE
int rsx_debug_sample_ablation_arm(bool on);      // rsx_sample.hip (development build only)
#endif

struct rsx_bpr_trainer {
    rsx_bpr_trainer_config c;
    int device = 0;
    hipStream_t side = nullptr;
    // the exchange of a sharded step issued by the library (RCCL, config.comm) and the range-by-range apply of the
    // chunked step run here: highest priority, so that their few workgroups get the next free wave slots under the step kernel
    hipStream_t aux = nullptr;
    // chunked + sharded, OPT-IN (rsx_set_option "apply_stream" = 1): the ranges' applies leave the collective stream, where
    // apply(k) holds back the collective of range k + 1 -- at the BASELINE configs[3] shape (1M items: 256 MB per range at two
    // ranges) an apply is ~150 us.  Measured (profiles/r04_exchange_model_config3.txt, us per step at an exchange of 0.5 / 1.0 /
    // 1.5 ms): on the collective stream 1154 / 1406 / 1898, on a stream of its own 1392 / 1537 / 1748 -- the fifth HIP stream costs
    // more than the serialisation until the exchange is longer than the step; off by default
    hipStream_t apply_st = nullptr;
    hipEvent_t ev_start = nullptr;               // run stream -> aux: the step's counters are reset, the tables are consistent
    hipEvent_t ev_g = nullptr;                   // run stream -> aux: G (folded) is complete
    hipEvent_t ev_x[2] = {};                     // aux -> run stream: the exchange (+ apply) of gradient buffer 0 / 1 is done
    // item chunks (config.chunks > 1): one stream per item range, in descending priority.  Range k's chain
    //   [all kernels of the step before done] -> kernel(k) -> (aux: fold, all-reduce of range k) -> apply(k)
    // lives on cs[k]; the next step's kernel(k) follows its own apply in stream order.
    hipStream_t cs[RSX_MAX_CHUNKS] = {};
    hipEvent_t ev_k[RSX_MAX_CHUNKS][2] = {};     // cs[k] -> everybody: kernel(k) of an even / odd step is done
    hipEvent_t ev_r[RSX_MAX_CHUNKS] = {};        // aux -> cs[k]: range k's rows of G are reduced over the ranks
    hipEvent_t ev_a[RSX_MAX_CHUNKS] = {};        // cs[k] -> run stream: apply(k) of the latest step is done
    bool kernels_in_flight = false;              // the kernels of the step before have been launched in this run
    // ring of RSX_TRAINER_SLOTS triplet buffers: the one being consumed and the batches sampled ahead.  Two
    // ahead, not one: with one, the step kernel's launch waits on an event the side stream has recorded only
    // microseconds before, and that cross-stream hand-over showed as a 11-12 us hole in front of EVERY step
    // kernel in the rocprofv3 time line (11 % of a 65 536-triplet step); an event that completed a whole step
    // earlier costs nothing.
    static constexpr int S = RSX_TRAINER_SLOTS;
    hipEvent_t ready[S] = {};                    // slot sampled (recorded on side)
    hipEvent_t freed[S] = {};                    // slot consumed (recorded on the run stream)
    hipEvent_t fork = nullptr;                   // run stream -> side ordering
    bool freed_valid[S] = {};
    int64_t step = 0;                            // next step to CONSUME
    int64_t epoch_pos = 0;                       // next position of the user permutation to SAMPLE
    int cur = 0;                                 // slot the next step consumes
    int ahead = 0;                               // slots cur, cur+1, ... (mod S) hold the batches of steps step, step+1, ...
    int64_t slot_batch[S] = {};
    int64_t slot_pos_before[S] = {};
    uint64_t slot_key[S] = {};
    int slot_nb[S] = {};
    bool slot_sorted[S] = {};                    // ordered by positive item without blocked negatives
    bool slot_chunked[S] = {};                   // sampled with the item-range rule (config.chunks > 1)
    int last = -1;                               // slot consumed by the most recent step
    bool flip = false;                           // stale_exchange: the next step accumulates into G_alt
    // live timing of the step kernel (HIP events on the stream the kernel is launched on, every `time_every`-th step).  A
    // chunked step has one kernel per item range, each on its own stream: one pair per range (`pairs` counts them, `timed`
    // the steps), and what is reported per step is the SUM of the ranges' kernel durations
    std::vector<hipEvent_t> t0, t1;
    size_t timed = 0, pairs = 0;
};

namespace {

int32_t *slot_ptr(const rsx_bpr_trainer *t, int slot, int which)
{
    return t->c.triplets + ((size_t)slot * 3 + which) * (size_t)t->c.batch;
}

// per-step key of the negative-block permutation (nonzero).  Same function as
// recsys_pytorch_amd/sharded.py:BPREngine._neg_key, so that a native run and hand-driven steps
// draw the same triplets.
uint64_t neg_key_for(uint64_t seed, int64_t step)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(step + 1) * 0xD1B54A32D192ED03ull;
    z ^= z >> 31;
    return z | 1ull;
}

// The trainer's events only order kernels of ONE device across its two streams; nobody on the host or on
// another device inspects memory through them.  Without hipEventDisableSystemFence every hipEventRecord on
// the run stream is a system-scope release -- a write-back of the L2s, which the apply sweep has just
// filled with 100 MB of dirty item rows -- and showed as an 11-12 us hole in front of every step kernel.
constexpr unsigned kOrderOnly = hipEventDisableTiming | hipEventDisableSystemFence;

bool chunked(const rsx_bpr_trainer *t) { return t->c.chunks > 1; }

int effective_neg_block(const rsx_bpr_trainer *t, int64_t batch)
{
    // (a chunked trainer lives in a relabelled item space with padding rows, which only the chunked sampler knows to
    //  avoid: every batch of it, whatever its size, takes the chunked layout)
    if (chunked(t)) return t->c.neg_block;
    return (t->c.neg_block > 0 && batch >= 2 * t->c.num_items) ? t->c.neg_block : 0;
}

int64_t *chunk_pos_ptr(const rsx_bpr_trainer *t, int slot) { return t->c.chunk_pos + (size_t)slot * (t->c.chunks + 1); }

#define RSX_TRY(call) do { int rc__ = (call); if (rc__ != RSX_OK) return rc__; } while (0)

// DEVELOPMENT BUILD ONLY (librsx_dev.so, -DRSX_ABLATE): a stand-in for the time an exchange takes on the wire.  On a one-GPU
// box the collectives of a one-rank communicator are identities; `rsx_debug_set_exchange_delay(us)` makes every FULL
// exchange of the item gradients hold the trainer's collective stream for `us` microseconds (a range's exchange for
// us / chunks), with eight idle-spinning workgroups -- the footprint of a collective kernel -- so that the schedules can be
// compared against an exchange of a given length (tools/exchange_model.py).  The shipped library
// has no such switch: there the function below is empty.
#ifdef RSX_ABLATE
static int g_exchange_delay_us = 0;
static int g_exchange_traffic = 0;
RSX_API int rsx_debug_set_exchange_delay(int us) { g_exchange_delay_us = us < 0 ? 0 : us; return RSX_OK; }
// `rsx_debug_set_exchange_traffic(1)`: the stand-in also MOVES the message while it holds the stream -- its eight workgroups read
// the exchanged rows and write the same values back (idempotent), paced over the delay: the HBM side of what an in-place all-reduce
// does to its own buffer (the peers' reads of it over xGMI come on top and are not modelled)
RSX_API int rsx_debug_set_exchange_traffic(int on) { g_exchange_traffic = on; return RSX_OK; }
__global__ __launch_bounds__(256) void exchange_delay_kernel(uint64_t ticks, float4 *buf, int64_t n4)
{
    const uint64_t t0 = wall_clock64();          // 100 MHz
    if (buf != nullptr) {
        // this workgroup's slice in 32 pieces, piece q not before q / 32 of the delay has passed; eight quads per thread in flight
        const int64_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * per, hi = (lo + per < n4) ? lo + per : n4;
        const int64_t piece = (per + 31) / 32;
        for (int q = 0; q < 32; ++q) {
            while (wall_clock64() - t0 < ticks * (uint64_t)q / 32) __builtin_amdgcn_s_sleep(8);
            const int64_t a = lo + q * piece, b = (a + piece < hi) ? a + piece : hi;
            for (int64_t e0 = a + threadIdx.x; e0 < b; e0 += 256 * 8) {
                float4 v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) if (e0 + c * 256 < b) v[c] = buf[e0 + c * 256];
#pragma unroll
                for (int c = 0; c < 8; ++c) if (e0 + c * 256 < b) { asm volatile("" : "+v"(v[c].x), "+v"(v[c].y), "+v"(v[c].z), "+v"(v[c].w)); buf[e0 + c * 256] = v[c]; }
            }
        }
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static int rsx_debug_exchange_delay(hipStream_t st, int parts, float *buf = nullptr, int64_t n = 0)
{
    if (g_exchange_delay_us <= 0) return RSX_OK;
    // (with traffic: 32 workgroups -- the channels a collective of this size runs on -- instead of 8)
    hipLaunchKernelGGL(exchange_delay_kernel, dim3(g_exchange_traffic ? 32 : 8), dim3(256), 0, st,
                       (uint64_t)g_exchange_delay_us * 100ull / (uint64_t)parts, g_exchange_traffic ? (float4 *)buf : (float4 *)nullptr, n / 4);
    return RSX_OK;
}
// `rsx_debug_set_sampler_replay(1)`: once every slot of the ring holds a batch, the sampler kernels are no longer launched and
// the loop steps on the three batches it has (with their own keys) -- the step WITHOUT a sampler beside it, to price what the
// sampler costs the loop (not training: the same 3 batches again and again).
// `rsx_debug_set_sampler_replay(2)`: the same replayed steps, but the sampler kernels DO run beside them, into a shadow buffer
// nobody reads -- so that variants of the sampler (rsx_debug_set_sample_ablation) disturb an IDENTICAL step workload.
static int g_sampler_replay = 0;
static int32_t *g_shadow = nullptr;
static int64_t g_shadow_n = 0;
RSX_API int rsx_debug_set_sampler_replay(int on) { g_sampler_replay = on; return RSX_OK; }
#else
static inline int rsx_debug_exchange_delay(hipStream_t, int, float * = nullptr, int64_t = 0) { return RSX_OK; }
#endif
#define RSX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            rsx_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__));       \
            return RSX_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

// queue the sampler of step `step_index` into `slot` on the side stream
int launch_sample(rsx_bpr_trainer *t, int slot, int64_t step_index, int64_t batch)
{
    const rsx_bpr_trainer_config &c = t->c;
    // a batch never straddles two passes over the user permutation (tail of a pass is dropped)
    t->slot_pos_before[slot] = t->epoch_pos;
    if ((t->epoch_pos % c.num_users) + batch > c.num_users) t->epoch_pos = (t->epoch_pos / c.num_users + 1) * c.num_users;
    const int nb = effective_neg_block(t, batch);
    const bool sorted = nb > 0 || (c.sort_min_batch > 0 && batch >= c.sort_min_batch);
    const uint64_t key = nb ? neg_key_for(c.seed_key, step_index) : 0ull;
    if (t->freed_valid[slot]) RSX_HIP(hipStreamWaitEvent(t->side, t->freed[slot], 0));
#ifdef RSX_ABLATE
    if (g_sampler_replay == 1 && step_index >= rsx_bpr_trainer::S && t->slot_batch[slot] == batch) {
        RSX_HIP(hipEventRecord(t->ready[slot], t->side));
        t->epoch_pos += batch;
        return RSX_OK;
    }
    if (g_sampler_replay == 2 && step_index >= rsx_bpr_trainer::S && t->slot_batch[slot] == batch && !chunked(t)) {
        if (g_shadow_n < 3 * batch) {
            if (g_shadow) (void)hipFree(g_shadow);
            RSX_HIP(hipMalloc((void **)&g_shadow, (size_t)3 * batch * sizeof(int32_t)));
            g_shadow_n = 3 * batch;
        }
        if (step_index == rsx_bpr_trainer::S) {      // the first shadow sample of this trainer: the whole samples before it are done
            RSX_HIP(hipStreamSynchronize(t->side));
            RSX_TRY(rsx_debug_sample_ablation_arm(true));
        }
        RSX_TRY(rsx_bpr_sample(c.indptr, c.indices, c.num_users, c.num_items, batch, c.seed, (uint64_t)step_index,
                               t->epoch_pos, nb, key, sorted ? RSX_SAMPLE_SORT_POS : 0u, sorted ? c.sample_ws : nullptr,
                               sorted ? c.sample_ws_bytes : 0, nb ? c.user_sig : nullptr, sorted ? c.item_cdf : nullptr,
                               g_shadow, g_shadow + batch, g_shadow + 2 * batch, (rsx_stream_t)t->side));
        RSX_HIP(hipEventRecord(t->ready[slot], t->side));
        t->epoch_pos += batch;
        return RSX_OK;
    }
#endif
    // a batch that holds every user once and wants an ordered layout: the CSC walk (no buckets, no sort; rsx_sample.hip)
    if (c.csc != nullptr && batch == c.num_users && t->epoch_pos % c.num_users == 0 && (chunked(t) || sorted))
        RSX_TRY(rsx_bpr_sample_csc(c.csc, c.indptr, c.indices, c.num_users, c.num_items, chunked(t) ? c.items_real : c.num_items,
                                   chunked(t) ? c.chunks : 1, c.seed, (uint64_t)step_index, nb, key, c.sample_ws, c.sample_ws_bytes,
                                   nb ? c.user_sig : nullptr, slot_ptr(t, slot, 0), slot_ptr(t, slot, 1), slot_ptr(t, slot, 2),
                                   chunked(t) ? chunk_pos_ptr(t, slot) : nullptr, (rsx_stream_t)t->side));
    else if (chunked(t))
        RSX_TRY(rsx_bpr_sample_chunked(c.indptr, c.indices, c.num_users, c.num_items, c.items_real, c.chunks, batch, c.seed,
                                       (uint64_t)step_index, t->epoch_pos, nb, key, c.sample_ws, c.sample_ws_bytes, c.user_sig,
                                       c.item_cdf, slot_ptr(t, slot, 0), slot_ptr(t, slot, 1), slot_ptr(t, slot, 2),
                                       chunk_pos_ptr(t, slot), (rsx_stream_t)t->side));
    else
    RSX_TRY(rsx_bpr_sample(c.indptr, c.indices, c.num_users, c.num_items, batch, c.seed, (uint64_t)step_index,
                           t->epoch_pos, nb, key, sorted ? RSX_SAMPLE_SORT_POS : 0u, sorted ? c.sample_ws : nullptr,
                           sorted ? c.sample_ws_bytes : 0, nb ? c.user_sig : nullptr, sorted ? c.item_cdf : nullptr,
                           slot_ptr(t, slot, 0), slot_ptr(t, slot, 1), slot_ptr(t, slot, 2), (rsx_stream_t)t->side));
    RSX_HIP(hipEventRecord(t->ready[slot], t->side));
    t->epoch_pos += batch;
    t->slot_batch[slot] = batch;
    t->slot_key[slot] = key;
    t->slot_nb[slot] = nb;
    t->slot_sorted[slot] = sorted && nb == 0;
    t->slot_chunked[slot] = chunked(t);
    return RSX_OK;
}

}  // namespace

RSX_API int rsx_bpr_trainer_create(const rsx_bpr_trainer_config *cfg, rsx_bpr_trainer **out)
{
    RSX_CHECK_ARG(cfg != nullptr && out != nullptr, "null pointer");
    RSX_CHECK_ARG(cfg->P && cfg->Q && cfg->G && cfg->indptr && cfg->indices && cfg->triplets, "null device pointer");
    RSX_CHECK_ARG(rsx_dim_ok(cfg->d), "d must be 32, 64, 128 or 256");
    RSX_CHECK_ARG(cfg->num_users > 0 && cfg->num_items > 0 && cfg->batch > 0 && cfg->batch <= cfg->num_users,
                  "batch must be in [1, num_users]");
    RSX_CHECK_ARG(cfg->neg_block >= 0 && cfg->neg_block <= kMaxNegBlock, "neg_block must be in [0, 16]");
    RSX_CHECK_ARG((cfg->neg_block == 0 && cfg->sort_min_batch <= 0) || (cfg->sample_ws != nullptr &&
                  cfg->sample_ws_bytes >= rsx_bpr_sample_workspace(cfg->batch, cfg->num_items)),
                  "neg_block / sort_min_batch need a sampler workspace of rsx_bpr_sample_workspace(batch, num_items) bytes");
    RSX_CHECK_ARG((cfg->hot_slot == nullptr) == (cfg->G_hot == nullptr) && (cfg->hot_slot == nullptr) == (cfg->hot_items == nullptr),
                  "hot_slot, G_hot and hot_items go together");
    RSX_CHECK_ARG((cfg->exchange_begin == nullptr) == (cfg->exchange_end == nullptr), "exchange_begin and exchange_end go together");
    RSX_CHECK_ARG(cfg->step0 >= 0 && cfg->epoch_pos0 >= 0, "negative start state");
    const bool native = cfg->comm != nullptr;
    RSX_CHECK_ARG(!(native && cfg->exchange_begin != nullptr), "give the exchange callbacks OR a communicator, not both");
    RSX_CHECK_ARG(!native || cfg->exchange_kind == RSX_EXCHANGE_ALLREDUCE || cfg->exchange_kind == RSX_EXCHANGE_SCATTER_GATHER,
                  "with comm: exchange_kind must be RSX_EXCHANGE_ALLREDUCE or RSX_EXCHANGE_SCATTER_GATHER");
    const bool meshed = cfg->mesh != nullptr;
    RSX_CHECK_ARG(!(meshed && (native || cfg->exchange_begin != nullptr || cfg->exchange_range != nullptr)),
                  "give ONE exchange: the callbacks, a communicator, or a mesh");
    RSX_CHECK_ARG(!(meshed && (cfg->two_pass || cfg->stale_exchange)), "a mesh exchanges and applies in one go: no two_pass / stale_exchange");
    if (cfg->csc != nullptr) {
        int64_t cnnz = 0;
        RSX_CHECK_ARG(rsx_csc_matches(cfg->csc, cfg->indptr, cfg->indices, cfg->num_users, cfg->num_items),
                      "csc was built from another CSR than this config's indptr / indices / num_users / num_items");
        RSX_TRY(rsx_csc_info(cfg->csc, &cnnz, nullptr, nullptr, nullptr));
        RSX_CHECK_ARG(cfg->sample_ws != nullptr && cfg->sample_ws_bytes >= rsx_bpr_sample_csc_workspace(cnnz),
                      "csc needs a sampler workspace of at least rsx_bpr_sample_csc_workspace(nnz) bytes");
    }
    if (meshed) {       // the mesh exchanges ITS OWN tables: they must be this trainer's
        const float *mQ = nullptr, *mG = nullptr;
        int64_t mrows = 0;
        int md = 0;
        rsx_mesh_tables(cfg->mesh, &mQ, &mG, &mrows, &md);
        RSX_CHECK_ARG(mQ == cfg->Q && mG == cfg->G && mrows >= cfg->num_items && md == cfg->d,
                      "the mesh was built over other tables than this trainer's Q / G (rsx_mesh_local takes the SAME Q, G, d and at least num_items rows)");
    }
    const bool sg = native && cfg->exchange_kind == RSX_EXCHANGE_SCATTER_GATHER;
    if (sg) {
        int world = 1;
        RSX_TRY(rsx_comm_info(cfg->comm, nullptr, &world));
        RSX_CHECK_ARG(cfg->item_rows_padded >= cfg->num_items && cfg->item_rows_padded % world == 0,
                      "RSX_EXCHANGE_SCATTER_GATHER: Q and G must hold item_rows_padded = world * shard rows");
    }
    RSX_CHECK_ARG(!cfg->stale_exchange || ((cfg->exchange_begin != nullptr || native) && !sg && cfg->G_alt != nullptr && cfg->G_alt != cfg->G),
                  "stale_exchange needs an exchange (callbacks or an all-reduce communicator) and a second gradient buffer G_alt");
    if (cfg->chunks > 1) {
        RSX_CHECK_ARG(cfg->chunks <= RSX_MAX_CHUNKS && cfg->item_cdf != nullptr && cfg->chunk_pos != nullptr &&
                      cfg->progress != nullptr && cfg->items_real > 0 && cfg->sample_ws != nullptr &&
                      cfg->sample_ws_bytes >= rsx_bpr_sample_workspace(cfg->batch, cfg->num_items),
                      "chunks > 1 needs item_cdf, chunk_pos, progress, items_real and the sampler workspace");
        RSX_CHECK_ARG(cfg->num_items == cfg->chunks * rsx_chunk_rows(cfg->items_real, cfg->chunks, cfg->neg_block),
                      "chunks > 1: num_items must be chunks * rsx_chunk_rows(items_real, chunks, neg_block)");
        RSX_CHECK_ARG(cfg->batch <= (1ll << 21), "chunks > 1: at most 2^21 triplets per step");
        RSX_CHECK_ARG(cfg->exchange_begin == nullptr && !sg && !cfg->stale_exchange && !cfg->two_pass,
                      "chunks > 1 replaces two_pass / stale_exchange; when sharded it takes the all-reduce communicator or exchange_range");
        RSX_CHECK_ARG(!(native && cfg->exchange_range != nullptr), "give exchange_range OR a communicator, not both");
    } else {
        RSX_CHECK_ARG(cfg->exchange_range == nullptr, "exchange_range is the per-range exchange of a chunked trainer (chunks > 1)");
    }
#ifdef RSX_ABLATE
    (void)hipDeviceSynchronize();
    (void)rsx_debug_sample_ablation_arm(false);      // a new trainer samples its first batches whole
#endif
    rsx_bpr_trainer *t = new (std::nothrow) rsx_bpr_trainer();
    if (t == nullptr) { rsx_set_error("rsx_bpr_trainer_create: out of memory"); return RSX_E_INVALID; }
    t->c = *cfg;
    t->step = cfg->step0;
    t->epoch_pos = cfg->epoch_pos0;
    // The sampler is the background producer: it has a whole step of slack, so its stream gets the LOWEST
    // priority and its workgroups fill the slots the step kernel leaves free (measured beside the 1M-triplet
    // step kernel: that kernel 348 -> 326 us, step 386 -> 372 us, same box; highest priority: no gain; round 3, beside the
    // pipelined kernel: default priority 349-359 vs 341-354 us per step; CUs of its own -- 16 / 32 / 64 of the 256 by CU mask,
    // the step on the others -- 757 / 439 / 391 vs 340: profiles/r03_exp_sampler_placement.txt)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);        // lo = numerically largest = least urgent
    const int prio = prio_lo;
    bool ok = hipGetDevice(&t->device) == hipSuccess &&
              hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, prio) == hipSuccess &&
              hipStreamCreateWithPriority(&t->aux, hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipEventCreateWithFlags(&t->ev_start, kOrderOnly) == hipSuccess &&
              hipEventCreateWithFlags(&t->ev_g, kOrderOnly) == hipSuccess &&
              hipEventCreateWithFlags(&t->ev_x[0], kOrderOnly) == hipSuccess &&
              hipEventCreateWithFlags(&t->ev_x[1], kOrderOnly) == hipSuccess &&
              hipEventCreateWithFlags(&t->fork, kOrderOnly) == hipSuccess;
    for (int s = 0; ok && s < rsx_bpr_trainer::S; ++s)
        ok = hipEventCreateWithFlags(&t->ready[s], kOrderOnly) == hipSuccess &&
             hipEventCreateWithFlags(&t->freed[s], kOrderOnly) == hipSuccess;
    if (ok && cfg->chunks > 1 && (native || cfg->exchange_range != nullptr)) {
        if (g_rsx_apply_stream == 1) ok = hipStreamCreateWithPriority(&t->apply_st, hipStreamNonBlocking, prio_hi) == hipSuccess;
    }
    for (int k = 0; ok && k < cfg->chunks && cfg->chunks > 1; ++k) {
        // range 0 most urgent, the others at the default priority; the LOWEST level stays with the sampler alone.  Measured
        // (one GPU, no exchange, us per step at C = 2 / 3 / 4): 443 / 467 / 550 with descending priorities, 433 / 474 / 684
        // all equal, 453 / 471 / 559 two urgent: the choice hardly matters, and one session with two range kernels plus
        // the sampler on the lowest level ran C = 4 at SECONDS per step (profiles/r03_exp_range_priorities.txt)
        const int pk = (k == 0) ? prio_hi : (prio_hi + 1 <= prio_lo - 1 ? prio_hi + 1 : prio_hi);
#if RSX_RANGE_ON_RUN_STREAM
        if (k == cfg->chunks - 1) t->cs[k] = nullptr; else
#endif
        ok = hipStreamCreateWithPriority(&t->cs[k], hipStreamNonBlocking, pk) == hipSuccess;
        ok = ok &&
             hipEventCreateWithFlags(&t->ev_k[k][0], kOrderOnly) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_k[k][1], kOrderOnly) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_r[k], kOrderOnly) == hipSuccess &&
             hipEventCreateWithFlags(&t->ev_a[k], kOrderOnly) == hipSuccess;
    }
    if (!ok) {
        rsx_set_error("rsx_bpr_trainer_create: could not create the side stream / events");
        rsx_bpr_trainer_destroy(t);
        return RSX_E_HIP;
    }
    *out = t;
    return RSX_OK;
}

RSX_API void rsx_bpr_trainer_destroy(rsx_bpr_trainer *t)
{
    if (t == nullptr) return;
    if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
    if (t->aux) { (void)hipStreamSynchronize(t->aux); (void)hipStreamDestroy(t->aux); }
    if (t->apply_st) { (void)hipStreamSynchronize(t->apply_st); (void)hipStreamDestroy(t->apply_st); }
    for (hipEvent_t e : {t->ev_start, t->ev_g, t->ev_x[0], t->ev_x[1]}) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < RSX_MAX_CHUNKS; ++k) {
        if (t->cs[k]) { (void)hipStreamSynchronize(t->cs[k]); (void)hipStreamDestroy(t->cs[k]); }
        for (hipEvent_t e : {t->ev_k[k][0], t->ev_k[k][1], t->ev_r[k], t->ev_a[k]}) if (e) (void)hipEventDestroy(e);
    }
    for (int s = 0; s < rsx_bpr_trainer::S; ++s) {
        if (t->ready[s]) (void)hipEventDestroy(t->ready[s]);
        if (t->freed[s]) (void)hipEventDestroy(t->freed[s]);
    }
    if (t->fork) (void)hipEventDestroy(t->fork);
    for (hipEvent_t e : t->t0) (void)hipEventDestroy(e);
    for (hipEvent_t e : t->t1) (void)hipEventDestroy(e);
    delete t;
}

RSX_API int rsx_bpr_trainer_run(rsx_bpr_trainer *t, int64_t n_steps, int64_t batch, int64_t global_batch,
                                int time_every, rsx_stream_t stream)
{
    RSX_CHECK_ARG(t != nullptr, "null trainer");
    RSX_CHECK_ARG(n_steps >= 0 && batch > 0 && batch <= t->c.batch, "batch must be in [1, config batch]");
    RSX_CHECK_ARG(global_batch >= batch, "global_batch is the sum of the ranks' batches");
    RSX_CHECK_ARG(time_every >= 0, "time_every must be >= 0");
    if (n_steps == 0) return RSX_OK;
    const rsx_bpr_trainer_config &c = t->c;
    hipStream_t st = (hipStream_t)stream;
    const float inv_batch = 1.0f / (float)global_batch;
    const bool native = c.comm != nullptr;                                     // the library issues the exchange itself (RCCL)
    const bool by_range = c.exchange_range != nullptr;                        // chunked: the caller's collective, range by range
    const bool meshed = c.mesh != nullptr;                                     // the library's own exchange over xGMI (rsx_mesh.hip)
    const bool sharded = c.exchange_begin != nullptr || native || by_range || meshed;
    const bool sg = native && c.exchange_kind == RSX_EXCHANGE_SCATTER_GATHER;
    const bool applies = sg || (!native && c.exchange_applies);                // the exchange leaves Q updated and G zero
    const bool hot = c.hot_slot != nullptr;
    const bool stale = sharded && c.stale_exchange != 0;
    t->timed = t->pairs = 0;
    constexpr int S = rsx_bpr_trainer::S;
    if (t->ahead > 0 && t->slot_batch[t->cur] != batch) {
        // batches of another size were sampled ahead (e.g. before an epoch's short last batch): hand their
        // positions back to the user permutation; the side stream is drained before the slots are reused
        t->epoch_pos = t->slot_pos_before[t->cur];
        t->ahead = 0;
        RSX_HIP(hipEventRecord(t->fork, t->side));
        RSX_HIP(hipStreamWaitEvent(st, t->fork, 0));
    }
    if (t->ahead == 0) {
        // the sampler reads only the CSR, but it must not start before earlier work of the run stream
        // that may still read the triplet buffers (a previous run's last step)
        RSX_HIP(hipEventRecord(t->fork, st));
        RSX_HIP(hipStreamWaitEvent(t->side, t->fork, 0));
        RSX_TRY(launch_sample(t, t->cur, t->step, batch));
        t->ahead = 1;
    }
    // keep the ring full: sample the batches of the steps after the one being consumed
    auto top_up = [&]() -> int {
        while (t->ahead < S) {
            RSX_TRY(launch_sample(t, (t->cur + t->ahead) % S, t->step + t->ahead, batch));
            ++t->ahead;
        }
        return RSX_OK;
    };
    // the one exchange of a step: the item gradients, summed over the ranks.  begin: `Gbuf` (folded) is complete on the
    // run stream, start the collective; end: the run stream waits for it.  Either the caller's callbacks (torch.distributed
    // in this package's CPU tests) or RCCL issued from here on the trainer's own stream.
    int my_rank = 0, world_size = 1;
    if (native) RSX_TRY(rsx_comm_info(c.comm, &my_rank, &world_size));
    auto exchange_begin = [&](float *Gbuf, int which) -> int {
        if (!native) {
            if (c.exchange_begin(c.exchange_ctx) != 0) { rsx_set_error("rsx_bpr_trainer_run: exchange_begin failed"); return RSX_E_INVALID; }
            return RSX_OK;
        }
        RSX_HIP(hipEventRecord(t->ev_g, st));
        RSX_HIP(hipStreamWaitEvent(t->aux, t->ev_g, 0));
        if (!sg) { RSX_TRY(rsx_comm_all_reduce(c.comm, Gbuf, c.num_items * c.d, t->aux)); RSX_TRY(rsx_debug_exchange_delay(t->aux, 1, Gbuf, c.num_items * c.d)); }
        else RSX_TRY(rsx_comm_reduce_scatter(c.comm, Gbuf, c.item_rows_padded / world_size * c.d, t->aux));
        RSX_HIP(hipEventRecord(t->ev_x[which], t->aux));
        return RSX_OK;
    };
    // end: the run stream waits for the exchange.  Scatter-gather: only now -- after everything of this step that READS
    // the item table (the user pass of a two-pass step) -- the rank applies ITS shard of item rows, zeroes the other
    // shards of G (they hold its partial sums) and all-gathers the updated Q rows: every row is computed by one rank
    // and copied to the others (replicas identical by construction).
    auto exchange_end = [&](float *Gbuf, int which) -> int {
        if (!native) {
            if (c.exchange_end(c.exchange_ctx) != 0) { rsx_set_error("rsx_bpr_trainer_run: exchange_end failed"); return RSX_E_INVALID; }
            return RSX_OK;
        }
        if (sg) {
            const int64_t shard = c.item_rows_padded / world_size, n = shard * c.d;
            RSX_HIP(hipEventRecord(t->ev_g, st));
            RSX_HIP(hipStreamWaitEvent(t->aux, t->ev_g, 0));
            RSX_TRY(rsx_apply_item_grad(c.Q + (size_t)my_rank * n, Gbuf + (size_t)my_rank * n, shard, c.d, c.lr, nullptr, nullptr, 0,
                                        (rsx_stream_t)t->aux));
            if (my_rank > 0) RSX_HIP(hipMemsetAsync(Gbuf, 0, (size_t)my_rank * n * sizeof(float), t->aux));
            if (my_rank + 1 < world_size)
                RSX_HIP(hipMemsetAsync(Gbuf + (size_t)(my_rank + 1) * n, 0, (size_t)(world_size - my_rank - 1) * n * sizeof(float), t->aux));
            RSX_TRY(rsx_comm_all_gather(c.comm, c.Q, n, t->aux));
            RSX_HIP(hipEventRecord(t->ev_x[which], t->aux));
        }
        RSX_HIP(hipStreamWaitEvent(st, t->ev_x[which], 0));
        return RSX_OK;
    };
    for (int64_t s = 0; s < n_steps; ++s) {
        const int cur = t->cur;
        RSX_HIP(hipStreamWaitEvent(st, t->ready[cur], 0));
#if !RSX_SAMPLER_BEHIND_KERNEL
        if (!sharded) RSX_TRY(top_up());                                       // beside this step's kernel
#endif
        const int32_t *u = slot_ptr(t, cur, 0), *i = slot_ptr(t, cur, 1), *j = slot_ptr(t, cur, 2);
        const int nb = t->slot_nb[cur];
        const uint64_t key = t->slot_key[cur];
        const bool timed = time_every > 0 && (s % time_every) == 0;
        auto time_begin = [&](hipStream_t on) -> int {
            if (!timed) return RSX_OK;
            if (t->pairs == t->t0.size()) {
                hipEvent_t a, b;      // timing events, but order-only like the others: no system-scope release per record
                RSX_HIP(hipEventCreateWithFlags(&a, hipEventDisableSystemFence));
                RSX_HIP(hipEventCreateWithFlags(&b, hipEventDisableSystemFence));
                t->t0.push_back(a); t->t1.push_back(b);
            }
            RSX_HIP(hipEventRecord(t->t0[t->pairs], on));
            return RSX_OK;
        };
        auto time_end = [&](hipStream_t on) -> int {
            if (timed) { RSX_HIP(hipEventRecord(t->t1[t->pairs], on)); ++t->pairs; }
            return RSX_OK;
        };
        if (t->slot_chunked[cur]) {
            // ---- the step as independent pipelines over item ranges (include/rsx.h: "item chunks") ------------------
            // cs[k]:  wait(sampler, every kernel of the step before) -> kernel(k) -> wait(all-reduce k) -> apply(k)
            // aux  :  for k in order: wait(kernel k) -> fold the range's hot rows -> all-reduce the range's rows of G
            // The run stream only joins: it frees the triplet slot when the kernels are done, and ends the run behind
            // the last applies.  Range k of the NEXT step follows apply(k) in cs[k]'s own order -- it does not wait for
            // the other ranges' exchanges, which travel under it.
            const ChunkGeom g = chunk_geom(c.items_real, c.chunks, c.neg_block < 1 ? 1 : c.neg_block);
            const int par = (int)(t->step & 1);
            if (s == 0) {        // the ranges' streams start behind whatever the run stream holds (a previous run, the caller's work)
                RSX_HIP(hipEventRecord(t->ev_start, st));
                for (int k = 0; k < c.chunks; ++k) RSX_HIP(hipStreamWaitEvent(RSX_RANGE_STREAM(k), t->ev_start, 0));
                t->kernels_in_flight = false;
            }
            if (sharded || RSX_SAMPLER_BEHIND_KERNEL) RSX_TRY(top_up());
            for (int k = 0; k < c.chunks; ++k) {
                hipStream_t ck = RSX_RANGE_STREAM(k);
                RSX_HIP(hipStreamWaitEvent(ck, t->ready[cur], 0));
                if (t->kernels_in_flight)      // user rows: every range's kernel of the step before has written its users
                    for (int q = 0; q < c.chunks; ++q)
                        if (q != k) RSX_HIP(hipStreamWaitEvent(ck, t->ev_k[q][1 - par], 0));
                RSX_TRY(time_begin(ck));       // each range's kernel on ITS stream: a pair per range, summed per step
                RSX_TRY(rsx_bpr_step_chunked(c.P, c.Q, c.G, c.num_users, c.num_items, c.items_real, c.chunks, u, i, j, batch, c.d, c.lr,
                                             inv_batch, c.loss_acc, c.hot_slot, c.G_hot, c.hot_replicas, nb, key, chunk_pos_ptr(t, cur),
                                             c.progress, k, 1, (rsx_stream_t)ck));
                RSX_TRY(time_end(ck));
                RSX_HIP(hipEventRecord(t->ev_k[k][par], ck));
            }
            for (int k = 0; k < c.chunks; ++k) {
                hipStream_t ck = RSX_RANGE_STREAM(k);
                const int64_t lo = (int64_t)k * g.Ic;
                float *Gk = c.G + (size_t)lo * c.d, *Qk = c.Q + (size_t)lo * c.d;
                if (meshed) {
                    // the library's own exchange: reduce-scatter by direct reads of the peers' rows -> the own slice applied ->
                    // all-gather of the updated rows from their owners (rsx_mesh.hip); it leaves Q updated and G zero
                    RSX_HIP(hipStreamWaitEvent(t->aux, t->ev_k[k][par], 0));
                    if (hot) RSX_TRY(rsx_fold_hot_grad_range(c.G, c.G_hot, c.hot_items, c.n_hot, c.hot_replicas, c.d, lo, lo + g.Ic, t->aux));
                    RSX_TRY(rsx_mesh_exchange_apply(c.mesh, lo, g.Ic, c.lr, (rsx_stream_t)t->aux));
                } else if (native || by_range) {
                    // one stream for every collective of the step, issued in range order on every rank: RCCL from here, or the
                    // caller's collective (exchange_range: it queues the all-reduce of the range's rows on that same stream)
                    RSX_HIP(hipStreamWaitEvent(t->aux, t->ev_k[k][par], 0));
                    if (hot) RSX_TRY(rsx_fold_hot_grad_range(c.G, c.G_hot, c.hot_items, c.n_hot, c.hot_replicas, c.d, lo, lo + g.Ic, t->aux));
                    if (native) RSX_TRY(rsx_comm_all_reduce(c.comm, Gk, g.Ic * c.d, t->aux));
                    else if (c.exchange_range(c.exchange_ctx, k, Gk, g.Ic * c.d, (rsx_stream_t)t->aux) != 0) {
                        rsx_set_error("rsx_bpr_trainer_run: exchange_range failed (range %d)", k);
                        return RSX_E_INVALID;
                    }
                    RSX_TRY(rsx_debug_exchange_delay(t->aux, c.chunks, Gk, g.Ic * c.d));
                    if (t->apply_st != nullptr) {       // the apply off the collective stream: the next range's collective follows at once
                        RSX_HIP(hipEventRecord(t->ev_r[k], t->aux));
                        RSX_HIP(hipStreamWaitEvent(t->apply_st, t->ev_r[k], 0));
                        RSX_TRY(rsx_apply_item_grad(Qk, Gk, g.Ic, c.d, c.lr, nullptr, nullptr, 0, (rsx_stream_t)t->apply_st));
                        RSX_HIP(hipEventRecord(t->ev_a[k], t->apply_st));
                        RSX_HIP(hipStreamWaitEvent(ck, t->ev_a[k], 0));
                        continue;
                    }
                    RSX_TRY(rsx_apply_item_grad(Qk, Gk, g.Ic, c.d, c.lr, nullptr, nullptr, 0, (rsx_stream_t)t->aux));
                } else {
                    RSX_HIP(hipStreamWaitEvent(t->aux, t->ev_k[k][par], 0));
                    RSX_TRY(rsx_apply_item_grad(Qk, Gk, g.Ic, c.d, c.lr, hot ? c.hot_slot + lo : nullptr, c.G_hot, c.hot_replicas,
                                                (rsx_stream_t)t->aux));
                }
                // (the apply is short and on the critical path of range k's next kernel: it runs on the highest-priority
                //  stream -- on the range's own, low-priority stream it waited ~250 us behind the next step's other kernels)
                RSX_HIP(hipEventRecord(t->ev_a[k], t->aux));
                RSX_HIP(hipStreamWaitEvent(ck, t->ev_a[k], 0));
            }
            t->kernels_in_flight = true;
            // the triplet slot is free when every range's kernel is done
            for (int k = 0; k < c.chunks; ++k) RSX_HIP(hipStreamWaitEvent(st, t->ev_k[k][par], 0));
            if (timed) ++t->timed;
            if (s + 1 == n_steps)      // a run ends behind its last applies
                for (int k = 0; k < c.chunks; ++k) RSX_HIP(hipStreamWaitEvent(st, t->ev_a[k], 0));
        } else {
        RSX_TRY(time_begin(st));
        const unsigned sorted_flag = t->slot_sorted[cur] ? RSX_BATCH_SORTED : 0u;
        const bool two_pass = sharded && c.two_pass && !stale;
        const unsigned f = RSX_USERS_UNIQUE | sorted_flag | (two_pass ? RSX_ITEMS_ONLY : 0u);
        // one-step-stale exchange: the steps of this trainer alternate between the two gradient buffers, G first
        float *const Gs = (stale && t->flip) ? c.G_alt : c.G;
        if (stale) t->flip = !t->flip;
        float *const Gprev = (Gs == c.G) ? c.G_alt : c.G;
        const int which = (Gs == c.G) ? 0 : 1;
        // small batches of an unsharded trainer: the plain kernel marks the rows of G it adds to and the apply visits those only
        // (2 B <= items: at most ~2/3 of the rows are touched; "touched_apply" = 2: wherever the plain kernel runs)
        uint8_t *const marks = (!sharded && c.touched != nullptr && nb == 0 && sorted_flag == 0u && !two_pass && g_rsx_touched_apply != 0 &&
                                (g_rsx_touched_apply == 2 || 2 * batch <= c.num_items)) ? c.touched : nullptr;
        RSX_TRY(rsx_bpr_step_ex(c.P, c.Q, Gs, c.num_users, c.num_items, u, i, j, batch, c.d, c.lr, inv_batch, c.loss_acc, f,
                                nullptr, 0, c.hot_slot, c.G_hot, c.hot_replicas, nb, key, marks, stream));
        RSX_TRY(time_end(st));
        if (timed) ++t->timed;
#if RSX_SAMPLER_BEHIND_KERNEL
        if (!sharded) {    // (development A/B) the sampler of step t + 2 starts when THIS kernel has ended: beside the apply
            RSX_HIP(hipEventRecord(t->fork, st));
            RSX_HIP(hipStreamWaitEvent(t->side, t->fork, 0));
            RSX_TRY(top_up());
        }
#endif
        if (!sharded) {
            if (marks != nullptr)
                RSX_TRY(rsx_apply_item_grad_touched(c.Q, c.G, c.num_items, c.d, c.lr, c.hot_slot, c.G_hot, c.hot_replicas, marks, st));
            else
            RSX_TRY(rsx_apply_item_grad_ex(c.Q, c.G, c.num_items, c.d, c.lr, c.hot_slot, c.G_hot, c.hot_replicas, batch >= c.num_items, st));
        } else if (meshed) {
            // one pass, the exchange exposed: the mesh sums, applies and redistributes the item rows (Q updated, G zero afterwards)
            if (hot) RSX_TRY(rsx_fold_hot_grad(c.G, c.G_hot, c.hot_items, c.n_hot, c.hot_replicas, c.d, stream));
            RSX_TRY(rsx_mesh_exchange_apply(c.mesh, 0, c.num_items, c.lr, stream));
            RSX_HIP(hipEventRecord(t->fork, st));
            RSX_HIP(hipStreamWaitEvent(t->side, t->fork, 0));
            RSX_TRY(top_up());
        } else {
            // the one exchange of the step: the item gradients, summed over the ranks.  It needs the
            // folded G; with two passes it travels under the user pass and the next step's sampler.
            if (hot) RSX_TRY(rsx_fold_hot_grad(Gs, c.G_hot, c.hot_items, c.n_hot, c.hot_replicas, c.d, stream));
            RSX_TRY(exchange_begin(Gs, which));
            RSX_HIP(hipEventRecord(t->fork, st));
            RSX_HIP(hipStreamWaitEvent(t->side, t->fork, 0));
            RSX_TRY(top_up());                                                 // beside the exchange
            if (stale) {
                // this step's exchange stays in flight under the NEXT step kernel; what is finished and applied
                // now is the exchange of the step before (nothing before the first step of a run)
                if (s > 0) {
                    RSX_TRY(exchange_end(Gprev, 1 - which));
                    if (!applies)
                        RSX_TRY(rsx_apply_item_grad(c.Q, Gprev, c.num_items, c.d, c.lr, nullptr, nullptr, 0, stream));
                }
                if (s + 1 == n_steps) {      // drain: a run leaves nothing unapplied
                    RSX_TRY(exchange_end(Gs, which));
                    if (!applies)
                        RSX_TRY(rsx_apply_item_grad(c.Q, Gs, c.num_items, c.d, c.lr, nullptr, nullptr, 0, stream));
                }
            } else {
                if (two_pass)
                    RSX_TRY(rsx_bpr_step(c.P, c.Q, c.G, c.num_users, c.num_items, u, i, j, batch, c.d, c.lr, inv_batch, nullptr,
                                         RSX_USERS_UNIQUE | RSX_USERS_ONLY | sorted_flag, nullptr, 0, c.hot_slot, c.G_hot, c.hot_replicas, nb,
                                         key, stream));
                RSX_TRY(exchange_end(Gs, which));
                if (!applies)
                    RSX_TRY(rsx_apply_item_grad(c.Q, c.G, c.num_items, c.d, c.lr, nullptr, nullptr, 0, stream));
            }
        }
        }
        RSX_HIP(hipEventRecord(t->freed[cur], st));
        t->freed_valid[cur] = true;
        t->last = cur;
        t->cur = (cur + 1) % S;
        --t->ahead;
        ++t->step;
    }
    return RSX_OK;
}

RSX_API int rsx_bpr_trainer_check(rsx_bpr_trainer *t, rsx_stream_t stream)
{
    RSX_CHECK_ARG(t != nullptr, "null trainer");
    if (t->c.chunks <= 1 || t->c.progress == nullptr) return RSX_OK;
    uint32_t h[RSX_PROGRESS_WORDS] = {};
    RSX_HIP(hipMemcpyAsync(h, t->c.progress, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    // (read and reset: a violation fails THIS check, not every later one of the trainer's life)
    RSX_HIP(hipMemsetAsync(t->c.progress + RSX_PROGRESS_VIOLATIONS, 0, sizeof(uint32_t), (hipStream_t)stream));
    RSX_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (h[RSX_PROGRESS_VIOLATIONS] != 0) {
        rsx_set_error("rsx_bpr_trainer_check: %u triplets touched an item row outside their range (they race with the other ranges' pipelines)",
                      h[RSX_PROGRESS_VIOLATIONS]);
        return RSX_E_INVALID;
    }
    return RSX_OK;
}

RSX_API int rsx_bpr_trainer_state(const rsx_bpr_trainer *t, int64_t *step, int64_t *epoch_pos)
{
    RSX_CHECK_ARG(t != nullptr, "null trainer");
    if (step) *step = t->step;
    // position the NEXT un-sampled batch starts from, as if nothing had been sampled ahead
    if (epoch_pos) *epoch_pos = t->ahead > 0 ? t->slot_pos_before[t->cur] : t->epoch_pos;
    return RSX_OK;
}

RSX_API int rsx_bpr_trainer_seek(rsx_bpr_trainer *t, int64_t step, int64_t epoch_pos, rsx_stream_t stream)
{
    RSX_CHECK_ARG(t != nullptr && step >= 0 && epoch_pos >= 0, "bad state");
    if (t->ahead > 0) {      // drop the batches sampled ahead; order the side stream before later work
        RSX_HIP(hipEventRecord(t->fork, t->side));
        RSX_HIP(hipStreamWaitEvent((hipStream_t)stream, t->fork, 0));
        t->ahead = 0;
    }
    t->step = step;
    t->epoch_pos = epoch_pos;
    return RSX_OK;
}

RSX_API int rsx_bpr_trainer_last_batch(const rsx_bpr_trainer *t, const int32_t **u, const int32_t **i,
                                       const int32_t **j, int64_t *batch, int *neg_block, uint64_t *neg_key)
{
    RSX_CHECK_ARG(t != nullptr, "null trainer");
    RSX_CHECK_ARG(t->last >= 0, "no step has run yet");
    if (u) *u = slot_ptr(t, t->last, 0);
    if (i) *i = slot_ptr(t, t->last, 1);
    if (j) *j = slot_ptr(t, t->last, 2);
    if (batch) *batch = t->slot_batch[t->last];
    if (neg_block) *neg_block = t->slot_nb[t->last];
    if (neg_key) *neg_key = t->slot_key[t->last];
    return RSX_OK;
}

RSX_API int rsx_bpr_trainer_kernel_ms(const rsx_bpr_trainer *t, double *mean_ms, int64_t *count)
{
    RSX_CHECK_ARG(t != nullptr && mean_ms != nullptr, "null pointer");
    double sum = 0.0;
    for (size_t k = 0; k < t->pairs; ++k) {
        float ms = 0.f;
        RSX_HIP(hipEventSynchronize(t->t1[k]));
        RSX_HIP(hipEventElapsedTime(&ms, t->t0[k], t->t1[k]));
        sum += ms;
    }
    *mean_ms = t->timed ? sum / (double)t->timed : 0.0;
    if (count) *count = (int64_t)t->timed;
    return RSX_OK;
}
