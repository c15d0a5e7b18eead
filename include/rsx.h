/*
 * include/rsx.h -- C ABI of librsx.so: the MI355X (gfx950) BPR-MF hot path.
 *
 * Drop-in boundary for yoongi0428/RecSys_PyTorch's MF model.  The reference
 * has no FFI for this path (its arithmetic is eager PyTorch); each entry point
 * below names the reference code it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the ABI.
 *   - every pointer named P, Q, G, *_dev, or documented "device" is a DEVICE
 *     pointer borrowed from the caller (kept alive and contiguous by it).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     launches are asynchronous on it.  Nothing here synchronises.
 *   - return value: 0 = ok, <0 = error (RSX_E_*); text via rsx_last_error()
 *     (thread-local).  No exceptions cross the ABI.
 *   - tables are row-major fp32: P [U x d], Q [I x d]; d in {32, 64, 128, 256}
 *     (the reference takes any hidden_dim, models/MF.py:19,23-24: the host side
 *     stores other widths with zero columns behind them, which stay zero);
 *     row indices are int32.
 */
#ifndef RSX_H
#define RSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSX_ABI_VERSION 8

#define RSX_OK 0
#define RSX_E_INVALID (-1)   /* bad argument (null pointer, unsupported d, K ...) */
#define RSX_E_HIP (-2)       /* a HIP runtime call failed                         */
#define RSX_E_WORKSPACE (-3) /* caller-provided workspace too small               */

/* flags for rsx_bpr_step */
#define RSX_USERS_UNIQUE 1u /* caller guarantees no user id repeats inside the batch  */
                            /* (true for rsx_bpr_sample output and for the reference's */
                            /* PairwiseGenerator, data/generators.py:182-195)          */

#define RSX_NO_UPDATE 2u    /* evaluate the loss only: P and G are not written             */
                            /* (models/MF.py:99-107 process_one_batch without backward)    */
/* One step in two passes over the same triplets (both need RSX_USERS_UNIQUE).  The item pass   */
/* fills G and the loss and leaves P alone; the user pass updates P and touches neither G nor    */
/* the loss.  Neither pass changes what the other reads (P, Q pre-step), so items-then-users    */
/* equals the single call bit for bit on the user side.  Purpose: when the step is sharded,      */
/* the all-reduce of G is launched after the item pass and travels under the user pass.          */
#define RSX_ITEMS_ONLY 4u
#define RSX_USERS_ONLY 8u
#define RSX_WIDE_OFFSETS 16u /* address rows with 64-bit offsets even when every table is below 4 GB (where  */
                             /* 32-bit offsets are used and are faster).  Tables of 4 GB and more always   */
                             /* take the 64-bit form; the flag exists so that form can be tested at any size */

#define RSX_DETERMINISTIC 32u /* bit-reproducible step for bisecting (needs RSX_USERS_UNIQUE and a workspace of   */
                              /* rsx_bpr_step_det_workspace bytes): no atomics; every item row sums its incidences */
                              /* in ascending batch position -- the order of autograd's index_add on the CPU --    */
                              /* after a stable device sort of the 2B (item, position) incidences; the loss is     */
                              /* summed by one workgroup in a fixed order.  Same step, to rounding, as the default */
                              /* path; slower (a debugging aid).  hot_slot_dev / neg_block are ignored.            */

#define RSX_BATCH_SORTED 64u  /* the batch is ordered by positive item (rsx_bpr_sample with RSX_SAMPLE_SORT_POS; needs  */
                              /* RSX_USERS_UNIQUE; ignored when neg_block > 0, which implies it).  Every lane group then */
                              /* walks a contiguous range of positions and sums runs of equal positive items in         */
                              /* registers: one update of G per run instead of one per triplet.  A hint, not a          */
                              /* contract: on an unordered batch the runs are simply of length one.                     */

/* flags for rsx_bpr_sample */
#define RSX_SAMPLE_SORT_POS 1u /* order the batch by positive item (needs a workspace)       */

#define RSX_LOSS_SLOTS 2048 /* the loss accumulator is float[RSX_LOSS_SLOTS]; the loss is the SUM of all its entries. */
                            /* Wavefronts add their partial sums to 64 entries kept one per 128-byte line (index   */
                            /* 32*s): atomics on one line serialise at ~40 per microsecond, and thousands of       */
                            /* wavefronts adding into 64 ADJACENT floats (two lines) cost a 65 536-triplet step    */
                            /* 45 of its 114 microseconds                                                          */

typedef void *rsx_stream_t;

typedef struct rsx_device_info {
    int device;
    int compute_units;
    int wavefront_size;
    int64_t total_mem_bytes;
    int lds_bytes_per_cu;
    int clock_khz;
    char arch[64];
} rsx_device_info;

/* ---- library -------------------------------------------------------------- */
int rsx_version(void);
const char *rsx_last_error(void);
int rsx_device_info_get(int device, rsx_device_info *out);

/* Process-wide options (the only mutable library state besides the per-device side streams of
 * rsx_score_topk).  Unknown names and out-of-range values return RSX_E_INVALID.
 *   "score_lanes"     1..4 (default 2): 8192-row passes of rsx_score_topk in flight, one HIP stream
 *                     each (the selection kernels of one pass run under the product of another)
 *   "sample_sort_cap" 0..2048 (default 0 = 2048): pairs a bucket of the sorted sampler may hold and
 *                     still be sorted in LDS; tests lower it to exercise the out-of-LDS path.  The
 *                     sampled triplets do not depend on it.
 *   "step_waves"      0 (default: the library's choice per row width) or 2..8: wavefronts per SIMD the blocked step kernel
 *                     (rsx_bpr_step with neg_block / RSX_BATCH_SORTED, rsx_bpr_step_chunked) may keep resident; the launch
 *                     reserves LDS accordingly.  The kernel shares the chip with the NEXT step's sampler (native loop): what it
 *                     does not occupy the sampler runs in.  The result does not depend on it.
 *   "apply_stream"    0 (default) / 1: a chunked, sharded trainer (item chunks with comm or exchange_range) runs the applies
 *                     of its item ranges on a high-priority stream of their own instead of behind the collectives on the
 *                     collective stream (pays only where an exchange is longer than the whole step).  Read at
 *                     rsx_bpr_trainer_create.  The result does not depend on it.
 *   "touched_apply"   0 / 1 (default) / 2: the native loop's row-marked apply (rsx_bpr_trainer_config.touched) never / where
 *                     2 * batch <= num_items / wherever the plain kernel runs.  The result does not depend on it.
 *   "mesh_blocks"     0 (default: one per CU) .. 4096: workgroups of the two kernels of rsx_mesh_exchange_apply (like a collective's
 *                     channels: over xGMI they are link-bound and a few dozen saturate the links; every workgroup more takes a wave slot
 *                     from the other item ranges' step kernels it runs beside).  The result does not depend on it.
 * There is no option that skips work: the development ablation switches of the kernels exist only
 * in the separate dev build (librsx_dev.so, -DRSX_ABLATE), never in librsx.so.                    */
int rsx_set_option(const char *name, int64_t value);

/* ---- BPR triplet step -------------------------------------------------------
 * Replaces, for the pairwise branch, one iteration of the reference loop
 *   models/MF.py:64-68   zero_grad / process_one_batch / backward / step
 *   models/MF.py:32-42   gather P[u], Q[i], Q[j]; r = sum(mul)
 *   models/MF.py:99-107  loss = -mean(log(sigmoid(r_pos - r_neg)))
 * with SGD as the optimizer.  Batch-synchronous like autograd: every gradient
 * is taken at the PRE-step tables, duplicates are summed.
 *
 * rsx_bpr_step  (phase 1 of a step)
 *   for each triplet b:  x = <P[u],Q[i]> - <P[u],Q[j]>,  g = -sigmoid(-x)*inv_batch
 *     item gradients   G[i] += g*P[u];  G[j] -= g*P[u]     (fp32 atomics, G is [I x d])
 *     user rows        P[u] -= lr * g * (Q[i]-Q[j])
 *         RSX_USERS_UNIQUE set : written in place by the owning wavefront
 *         otherwise            : summed per distinct user in `ws`, then applied
 *                                (exact for repeated users)
 *   loss_acc (nullable): float[RSX_LOSS_SLOTS]; sum_b softplus(-x_b) is ADDED
 *                        (= the reference's -log(sigmoid(x_b)), models/MF.py:105, in the form that does not overflow: at
 *                        x_b < -88.7 the reference's fp32 sigmoid is 0 and its loss +inf -- the gradient, -sigmoid(-x) / B,
 *                        is finite and the same either way; this library reports the finite value -x_b),
 *     spread over the entries (loss of the batch = sum(all entries) * inv_batch).
 *   inv_batch = 1 / (global batch size)  (the mean of MF.py:105; with user
 *     sharding it is 1/(sum over ranks))
 *   ws / ws_bytes: device scratch, needed only without RSX_USERS_UNIQUE;
 *     size from rsx_bpr_step_workspace().  `ws` must be zero-filled once by
 *     the caller before first use; the call leaves it zero-filled again.
 *   Q is NOT modified here.  G must be zero before the first step; it is
 *   consumed (and re-zeroed) by rsx_apply_item_grad.
 *   Triplets with i < 0 are skipped (users without positives).
 *   hot_slot_dev / G_hot / hot_replicas (all nullable/0): contention relief for very
 *     popular items.  hot_slot_dev is int32 [num_items], the hot slot s of an item or -1;
 *     the positive-item gradient of a hot item goes to one of `hot_replicas` (power of
 *     two) private rows G_hot[(s*hot_replicas + r)*d ...] instead of G[i].  The caller
 *     runs rsx_fold_hot_grad before G is all-reduced / applied.  Sums are unchanged.
 *   neg_block (0 = off, <= 16; needs RSX_USERS_UNIQUE): the batch was drawn by
 *     rsx_bpr_sample with the same neg_block, i.e. the negative of triplet b lies in the
 *     item block floor(floor(b*I/B)/neg_block).  One wavefront then owns all triplets of
 *     a block and sums their negative-side gradients in LDS before touching G (one row
 *     update per item instead of one per triplet).  neg_key is the sampler's block
 *     permutation key (0 = identity).  Consecutive triplets with the same positive item
 *     (RSX_SAMPLE_SORT_POS) are summed in registers and reach G once per run.  Triplets
 *     that do not honour either contract are still summed correctly (atomic path).
 *     With hot_slot_dev the runs of popular items are flushed into the replicas.
 */
int64_t rsx_bpr_step_workspace(int64_t num_users, int64_t max_batch, int d);
int64_t rsx_bpr_step_det_workspace(int64_t batch, int64_t num_items);   /* scratch of RSX_DETERMINISTIC */

int rsx_bpr_step(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items,
                 const int32_t *u_dev, const int32_t *i_dev, const int32_t *j_dev, int64_t batch,
                 int d, float lr, float inv_batch, float *loss_acc, unsigned flags,
                 void *ws, int64_t ws_bytes, const int32_t *hot_slot_dev, float *G_hot,
                 int hot_replicas, int neg_block, uint64_t neg_key, rsx_stream_t stream);

/* G[hot_items[s]] += sum_r G_hot[s][r];  G_hot = 0.   hot_items_dev: int32 [n_hot].      */
int rsx_fold_hot_grad(float *G, float *G_hot, const int32_t *hot_items_dev, int n_hot,
                      int hot_replicas, int d, rsx_stream_t stream);

/* rsx_apply_item_grad  (phase 2 of a step; after the all-reduce of G when sharded)
 *   Q -= lr * G ;  G = 0        for every row of the [num_items x d] tables.
 *   Replaces the item-table half of optimizer.step() (models/MF.py:68) for SGD.
 *   Rows whose gradient is entirely zero are not written.
 *   hot_slot_dev / G_hot / hot_replicas (nullable): fold the popular rows' replicas in
 *   here instead of calling rsx_fold_hot_grad (only when no all-reduce of G sits between
 *   rsx_bpr_step and this call: the all-reduce needs the folded G).                 */
int rsx_apply_item_grad(float *Q, float *G, int64_t num_items, int d, float lr,
                        const int32_t *hot_slot_dev, float *G_hot, int hot_replicas,
                        rsx_stream_t stream);

/* ---- reference-exact optimizer ----------------------------
 * The reference ships dense Adam (models/MF.py:30).  rsx_bpr_grad leaves the tables
 * untouched and sums the DENSE gradients of the batch (loss.backward(), MF.py:67; duplicates
 * summed, any triplets): GP [num_users x d] += dP, GQ [num_items x d] += dQ, both zero
 * between steps.  rsx_adam_apply then is torch.optim.Adam's single-tensor update over ALL n
 * elements of one table (rows with a zero gradient move too once their moments are non-zero):
 *   m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;  w -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * and zeroes G.  t = 1 for the first step.  Call it once per table per step.
 * lr, the betas and eps are DOUBLES, as torch holds them (Python floats): the reference's kernel receives 1 - beta2 = 0.001
 * rounded ONCE to fp32 (ABI 7; with float arguments the library could only form 1.0f - 0.999f = 0.00099998713, 1.3e-5 off:
 * found by the round-5 fuzz campaign, which checks the step against torch's formula on the device's own p, m, v, g).    */
int rsx_bpr_grad(const float *P, const float *Q, float *GP, float *GQ, int64_t num_users,
                 int64_t num_items, const int32_t *u_dev, const int32_t *i_dev, const int32_t *j_dev,
                 int64_t batch, int d, float inv_batch, float *loss_acc, rsx_stream_t stream);
int rsx_adam_apply(float *W, float *M, float *V, float *G, int64_t n, double lr, double beta1,
                   double beta2, double eps, int64_t t, rsx_stream_t stream);

/* rsx_pointwise_grad: the POINTWISE branch of the reference model (models/MF.py:99-102 with hparams['pointwise'] = True;
 * "widening" row beyond SURVEY section 8f).  Batch of n (user, item, rating) entries -- users and items repeat, as in the
 * batches of data/generators.py:105-130 -- x_b = <P[u_b], Q[i_b]>, loss = mean_b l(x_b, y_b) with
 *   loss_kind 0: F.binary_cross_entropy_with_logits (hparams['loss_func'] != 'mse', MF.py:21)   dl/dx = sigmoid(x) - y
 *   loss_kind 1: F.mse_loss                                                                      dl/dx = 2 (x - y)
 * Dense gradients like rsx_bpr_grad: GP[u_b] += dl/dx * inv_n * Q[i_b], GQ[i_b] += dl/dx * inv_n * P[u_b] (tables
 * untouched; both buffers zero between steps); loss_acc (nullable, RSX_LOSS_SLOTS floats) += sum_b l_b, striped.
 * GP = GQ = NULL (with loss_acc): the loss alone, nothing written but loss_acc (process_one_batch, MF.py:99-102).
 * Then rsx_adam_apply on both tables (the optimizer as shipped, MF.py:30) or rsx_apply_item_grad on both (SGD). */
int rsx_pointwise_grad(const float *P, const float *Q, float *GP, float *GQ, int64_t num_users, int64_t num_items,
                       const int32_t *u_dev, const int32_t *i_dev, const float *y_dev, int64_t n, int d,
                       float inv_n, int loss_kind, float *loss_acc, rsx_stream_t stream);

/* rsx_pair_score: r[b] = <P[u[b]], Q[i[b]]>   (models/MF.py:38-42, MF.forward)            */
int rsx_pair_score(const float *P, const float *Q, const int32_t *u_dev, const int32_t *i_dev,
                   int64_t n, int d, float *r_out, rsx_stream_t stream);

/* ---- on-device triplet sampler ------------------------------------------------
 * Replaces data/generators.py:151-224 (PairwiseGenerator: host-side numpy
 * sampling + permutation + H2D copy).  Sampling semantics are BPR's, not the
 * reference's quirks (SURVEY appendix A, Q1-Q3 documented in DESIGN.md):
 *   user  : position (epoch_pos + b) of a keyed pseudo-random PERMUTATION of the
 *           local users -> no user repeats while batch <= num_users
 *           (the reference also visits each user once per epoch, generators.py:206-210);
 *           a batch of exactly num_users positions starting a pass holds every user once
 *           and is walked in id order (a batch is a set; the reads become coalesced)
 *   pos i : uniform over the user's CSR row (indices[indptr[u]:indptr[u+1]])
 *   neg j : uniform over [0,num_items) rejected while j is in that row
 *           (generators.py:178-185: p = 0 on the user's positives)
 * RNG: counter-based (seed, step, b) -> splitmix64 -> xorshift32 stream; the
 * result does not depend on launch geometry.  Users with an empty row get i=-1.
 * neg_block = c > 0: the negative of batch position p is drawn uniformly from the item
 *   block pi(w)*c .. pi(w)*c + c, w = floor(floor(p*I/B)/c), pi = permutation of the
 *   blocks keyed by neg_key (0 = identity), instead of from all items.  A user's
 *   position in the keyed user permutation is uniform, so each user still sees a
 *   uniformly distributed negative (exactly so when c | I and I | B); what changes is
 *   that the negatives of one step are stratified over the catalog (B/I per item on
 *   average) rather than independent.  See rsx_bpr_step(neg_block).
 * flags & RSX_SAMPLE_SORT_POS: the (user, positive) pairs are sorted by positive item
 *   before negatives are drawn per sorted position (pass a fresh nonzero neg_key per
 *   step so that a positive item's rank does not pin the negative block).  Needs
 *   ws of rsx_bpr_sample_workspace() bytes (scratch, contents irrelevant).
 * indptr: int64 [num_users+1], indices: int32, sorted within each row.          */
int64_t rsx_bpr_sample_workspace(int64_t batch, int64_t num_items);

int rsx_bpr_sample(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                   int64_t num_items, int64_t batch, uint64_t seed, uint64_t step,
                   int64_t epoch_pos, int neg_block, uint64_t neg_key, unsigned flags,
                   void *ws, int64_t ws_bytes, const uint64_t *user_sig_dev,
                   const uint32_t *item_cdf_dev, int32_t *u_out, int32_t *i_out, int32_t *j_out,
                   rsx_stream_t stream);

/* Optional accelerator of the rejection test of the sorted layout: user_sig_dev holds one
 * 16-byte record per user (two uint64, 16-byte aligned; static per CSR and neg_block, built by
 * rsx_bpr_build_signature; sig_out = 16 * num_users bytes).  Word 0 has one hashed bit per item
 * block that holds a positive of u: a clear bit proves every item of the drawn block negative
 * for u, so indptr and the row are not read at all for that draw.  Word 1 is the row's start in
 * `indices` (low 40 bits) and its length (high 24 bits), which saves the indptr look-up when the
 * row does have to be read.  NULL = always check the row through indptr.  Same draws either way. */
int rsx_bpr_build_signature(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                            int neg_block, uint64_t *sig_out, rsx_stream_t stream);

/* Optional accelerator of RSX_SAMPLE_SORT_POS: item_cdf_dev (uint32 [num_items + 1], static per
 * CSR) is the CDF of the positive-item distribution (item i with weight sum over its users of
 * 1/deg(u)) in 2^-32 units.  With it the batch is ordered by positive item without a device-wide
 * sort: the pairs are cut into item-range buckets of equal expected size from the CDF (a popular
 * item spreads over several buckets by a hash of the user) and every bucket is sorted by
 * (item, user) inside one workgroup; batches above 2^21 positions are ordered piece by piece.
 * NULL = device radix sort by item.  Same triplet distribution either way; the order inside the
 * batch differs, and is in both cases a pure function of (seed, step, CSR).
 * ws for the build: rsx_bpr_item_cdf_workspace(num_items) bytes.                                 */
int64_t rsx_bpr_item_cdf_workspace(int64_t num_items);
int rsx_bpr_build_item_cdf(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                           int64_t num_items, uint32_t *cdf_out, void *ws, int64_t ws_bytes,
                           rsx_stream_t stream);

/* ---- item chunks: the step as independent pipelines over item ranges ------------------------------
 * Purpose: the exchange of a sharded step needs the item gradients G COMPLETE, and the next step needs
 * the updated item table, so an exchange cannot slide under the neighbouring step kernels -- unless the step is cut so
 * that a part of the triplets touches a part of the item rows ONLY.  With `chunks` = C > 1 the item rows are cut into C
 * contiguous ranges of chunk_rows rows and
 *   - the batch is ordered by positive item (RSX_SAMPLE_SORT_POS), so the positions whose positive lies in range k
 *     are contiguous: [chunk_pos[k], chunk_pos[k+1]);
 *   - the negative of such a position is drawn from an item block of the SAME range (the block permutation is keyed
 *     per step and per range), so every triplet of those positions reads and writes item rows of range k only.
 * Range k of a step is then its own pipeline  step kernel(k) -> [fold hot rows, all-reduce rows of G] -> apply(k)  which
 * depends on NOTHING of the other ranges' item rows: range k of step t+1 may start as soon as range k of step t has
 * been exchanged and applied (and every range's kernel of step t is done: the user rows), while the exchanges of the
 * other ranges of step t are still travelling.  Synchronous semantics -- every row a triplet reads carries ALL updates
 * of the steps before -- with the exchange under the neighbouring kernels.  The native loop launches each range's
 * kernel on a stream of its own, in descending priority, so that the ranges finish staggered.
 * The caller makes the ranges statistically alike by training on a RELABELLED item space (a fixed permutation of the
 * item ids that balances the ranges' sampling mass, redrawn between native runs every few dozen steps; recsys_pytorch_amd/sharded.py:
 * BPREngine.set_chunks): the library sees item ids 0 .. C * chunk_rows, of which range k holds real(k) = items_real / C
 * (+1 for k < items_real % C) real items at its start and padding rows (never sampled, gradient always zero) behind.
 * Sampling semantics: a user's negative is uniform over the real items of the range its sampled positive fell in --
 * with the relabelling a pseudo-random 1/C of the catalog per positive item, another one after every redraw (over a fit
 * every item meets every other as a negative).
 * Sums are exactly those of rsx_bpr_step on the same triplets (tests replay the dumped triplets through the oracle).
 * Contract: triplets that do not honour the range rule are still summed, but race with the other ranges' pipelines;
 * the kernel counts them in progress[RSX_PROGRESS_VIOLATIONS] and the native loop reports the run failed.
 *
 * WITHOUT BLOCKS (neg_block = 0; batches below two triplets per item, where a block has nothing to sum on chip -- the shape of
 * BASELINE configs[3], 1.25M triplets on 1M items per GPU): the same ranges and the same pipelines; the negative of a position is
 * uniform over the REAL items of its positive's range (rejecting the user's own), the step walks the ordered batch like
 * RSX_BATCH_SORTED (positive runs summed in registers, negatives to G one by one), one launch per range whose wavefronts split
 * the range's positions evenly.  Everything below accepts neg_block = 0 for this form.
 *
 * rsx_chunk_rows: rows per range = ceil(items_real / chunks) rounded up to a multiple of neg_block (of 1 for neg_block = 0).
 * rsx_bpr_sample_chunked: rsx_bpr_sample(RSX_SAMPLE_SORT_POS, item_cdf) over the relabelled CSR with the range rule;
 *   num_items = chunks * chunk_rows; item_cdf_dev built over those ids (padding rows have no mass); batch <= 2^21;
 *   chunk_pos_out: int64 [chunks + 1] device, first batch position of every range (+ number of live positions).
 *   A user whose row covers its whole range has no negative there and is skipped (i = j = -1 at its position); one whose
 *   row covers all but a fraction f of the range is skipped with probability (1 - f)^128 (192 draws, never from elsewhere).
 * rsx_bpr_step_chunked: the blocked step kernel (rsx_bpr_step with neg_block, RSX_USERS_UNIQUE) over the positions of the
 *   ranges [first_range, first_range + num_ranges); progress: uint32 [RSX_PROGRESS_WORDS] device, zeroed by the caller ONCE (at allocation);
 *   progress[RSX_PROGRESS_VIOLATIONS] counts the triplets that left their range.                                  */
#define RSX_MAX_CHUNKS 8
#define RSX_PROGRESS_WORDS 16
#define RSX_PROGRESS_VIOLATIONS 8     /* triplets outside their range (must stay 0)                       */
int64_t rsx_chunk_rows(int64_t items_real, int chunks, int neg_block);

int rsx_bpr_sample_chunked(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                           int64_t num_items, int64_t items_real, int chunks, int64_t batch, uint64_t seed,
                           uint64_t step, int64_t epoch_pos, int neg_block, uint64_t neg_key, void *ws,
                           int64_t ws_bytes, const uint64_t *user_sig_dev, const uint32_t *item_cdf_dev,
                           int32_t *u_out, int32_t *i_out, int32_t *j_out, int64_t *chunk_pos_out,
                           rsx_stream_t stream);

int rsx_bpr_step_chunked(float *P, const float *Q, float *G, int64_t num_users, int64_t num_items,
                         int64_t items_real, int chunks, const int32_t *u_dev, const int32_t *i_dev,
                         const int32_t *j_dev, int64_t batch, int d, float lr, float inv_batch, float *loss_acc,
                         const int32_t *hot_slot_dev, float *G_hot, int hot_replicas, int neg_block,
                         uint64_t neg_key, const int64_t *chunk_pos_dev, uint32_t *progress_dev,
                         int first_range, int num_ranges, rsx_stream_t stream);

/* ---- whole-pass batches: the CSC walk -------------------------------------------------------------
 * Replaces data/generators.py:151-224 for a batch that holds EVERY user once (batch == num_users -- the reference's epoch: one
 * triplet per user, generators.py:206-210; the shape of every BASELINE config's bench step).  Such a batch needs no user
 * permutation, and its order by positive item is the order of the TRANSPOSED interaction matrix: rsx_bpr_build_csc writes, once per
 * CSR, the entries item by item (users ascending) as (user uint32, rank of the item in the user's row | row length) -- 6 bytes per
 * interaction, 8 when a row is longer than 255 (rows up to 65535) -- and rsx_bpr_sample_csc streams them once per step:
 *   pos i : entry e is kept iff  floor(h(seed, step, u) * deg(u) / 2^32) == rank(e): one positive per user, uniform in its row
 *           (h: a keyed 32-bit integer hash of the user id; users with an empty row or deg >= num_items get no triplet);
 *   order : the kept (user, item) pairs are compacted IN ORDER (wavefront ballots, a workgroup scan, decoupled look-back between
 *           the tiles): the batch comes out ordered by (item, user) without bucketing or sorting;
 *   neg j : per ordered position exactly as rsx_bpr_sample(RSX_SAMPLE_SORT_POS) / rsx_bpr_sample_chunked draw it (neg_block,
 *           neg_key, user_sig, the item ranges with chunks > 1 and chunk_pos_out): same rule, same rejection test.
 * Positions [n_live, num_users) (users without a triplet) carry i = j = -1.  The triplet DISTRIBUTION is that of the bucket path; the
 * draws differ (the positive is keyed by the user id instead of by the batch position).
 *   rsx_bpr_csc_bytes / _workspace   device bytes of the blob (the caller's, kept alive as long as the handle) / of the build scratch
 *   rsx_bpr_build_csc                a set-up call: it waits for `stream` once (the longest row decides the entry format)
 *   rsx_csc_destroy                  frees the host-side handle only
 *   rsx_bpr_sample_csc_workspace     scratch of one sampling call (tile states of the look-back; cleared by the call itself)     */
typedef struct rsx_csc rsx_csc;
int64_t rsx_bpr_csc_bytes(int64_t nnz, int64_t num_items);
int64_t rsx_bpr_csc_workspace(int64_t nnz, int64_t num_items);
int rsx_bpr_build_csc(const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users, int64_t num_items, int64_t nnz,
                      void *blob_dev, int64_t blob_bytes, void *ws, int64_t ws_bytes, rsx_stream_t stream, rsx_csc **out);
void rsx_csc_destroy(rsx_csc *c);
int rsx_csc_info(const rsx_csc *c, int64_t *nnz, int64_t *num_items, int *entry_bytes, int64_t *tiles);
int64_t rsx_bpr_sample_csc_workspace(int64_t nnz);
int rsx_bpr_sample_csc(const rsx_csc *csc, const int64_t *indptr_dev, const int32_t *indices_dev, int64_t num_users,
                       int64_t num_items, int64_t items_real, int chunks, uint64_t seed, uint64_t step, int neg_block,
                       uint64_t neg_key, void *ws, int64_t ws_bytes, const uint64_t *user_sig_dev, int32_t *u_out,
                       int32_t *i_out, int32_t *j_out, int64_t *chunk_pos_out, rsx_stream_t stream);

/* ---- RCCL from the library --------------------------------------------------------------------
 * One process per GPU, one communicator per process (created on the current device).  RCCL is bound at run time
 * (dlopen librccl.so.1: the copy torch.distributed already loaded, else ROCm's).  rank 0 calls rsx_comm_unique_id and
 * hands the RSX_COMM_ID_BYTES to every rank through the host's bootstrap (torch.distributed's store here); all ranks
 * then call rsx_comm_create (collective).  The communicator is given to rsx_bpr_trainer_create, which issues the
 * step's exchange itself: no callback into the interpreter inside the timed region.                      */
#define RSX_COMM_ID_BYTES 128
typedef struct rsx_comm rsx_comm;
int rsx_comm_unique_id(void *id_out);
int rsx_comm_create(const void *id, int rank, int world, rsx_comm **out);
void rsx_comm_destroy(rsx_comm *c);
int rsx_comm_info(const rsx_comm *c, int *rank, int *world);
int rsx_comm_all_reduce_f32(rsx_comm *c, float *buf_dev, int64_t n, rsx_stream_t stream);   /* sum, in place */

/* ---- the library's own exchange: a direct full mesh over xGMI ---------------------------------
 * "reduce-scatter + all-gather with all 7 peers concurrently rather than a ring".  The reference has
 * no multi-device code (main.py:24-27); what is preserved is its batch-synchronous step (models/MF.py:64-68).
 * Every rank maps the peers' Q, G and a mailbox (hipIpcOpenMemHandle); the rows of one exchange are cut into `world` slices;
 * rank r sums ITS slice of G by reading the peers' partial sums directly (N - 1 links at once), applies it to its slice of Q,
 * and every rank then copies the other slices' updated rows from their owners.  Every row is computed by one rank and copied:
 * replicas identical by construction.  Ordering between the ranks is device side (sequence-numbered flags, system-scope
 * release / acquire); no host thread takes part once the launches are queued.
 *   rsx_mesh_local     allocates the rank's mailbox, describes its tables -> desc_out (RSX_MESH_DESC_BYTES).  Q and G
 *                      [rows x d] are BORROWED and must stay allocated until rsx_mesh_destroy on EVERY rank.  What is exported
 *                      is VALIDATED first: each table must be plain device memory of the current device lying inside ONE
 *                      allocation (hipPointerGetAttributes, hipMemGetAddressRange); an allocation that holds both tables is
 *                      exported once.  Every HIP call that can refuse reports itself with its arguments (which table, base,
 *                      size, offset).  hipIpcGetMemHandle is retried a bounded number of times (the runtime may refuse while
 *                      peers still detach from an EARLIER export of the same allocation -- which the caller's barrier after
 *                      rsx_mesh_destroy rules out); rsx_mesh_export_retries says how many calls failed before the
 *                      exports succeeded (0 normally).
 *   rsx_mesh_connect   after the caller has gathered all ranks' descriptors in rank order (host bootstrap, e.g.
 *                      torch.distributed.all_gather_object): opens the peers' buffers.  world <= 16.
 *   rsx_mesh_exchange_apply(first_row, rows, lr)   collective in the sense that every rank must queue the same sequence
 *                      of calls: afterwards (stream order) Q[first_row .. +rows) -= lr * sum over the ranks of G[...] on
 *                      every rank, and those rows of G are zero.  What precedes it on `stream` must have completed G.
 *                      The exchanges of ONE mesh must execute in the order they were queued -- one stream, or streams
 *                      ordered by events: a flag holds the LATEST sequence number, so an exchange overtaking an earlier
 *                      one would release the peers' waits for both.
 *   rsx_mesh_check     synchronises `stream`; fails if a wait for a peer's signal gave up (rsx_mesh_set_wait_limit, default
 *                      20 s): such an exchange leaves wrong rows and says so -- it never hangs the GPU.
 *   rsx_mesh_destroy   the caller makes sure (host barrier) that no peer still reads this rank's buffers, and -- before any
 *                      rank exports the same allocations again -- that every rank has returned from rsx_mesh_destroy
 *                      (a second host barrier: recsys_pytorch_amd/rsx.py Mesh.close).
 * Handed to rsx_bpr_trainer_create as config.mesh, the native loop issues it per step or per item range.
 *   rsx_mesh_alloc     device memory FOR the exchanged tables: plain hipMalloc'ed, zero-filled, and exported at allocation -- an
 *                      allocation the runtime refuses to export is set aside and another one taken (up to 8), so what comes back HAS
 *                      an IPC handle, which the library keeps: rsx_mesh_local over such memory makes no export call at all.  Twice (the
 *                      driver's round-5 box, a round-6 box) the runtime refused to export a POOLED allocation of the caller's allocator
 *                      -- persistently, "invalid argument" -- in a process that had mapped and unmapped peers' memory before; tables that
 *                      live in rsx_mesh_alloc memory cannot meet that.  rsx_mesh_free (after rsx_mesh_destroy on every rank) hands the
 *                      block back to the LIBRARY, which keeps it -- allocation and handle -- for the next rsx_mesh_alloc of that size:
 *                      an exporter that really frees memory its peers had mapped and gets the same address again hands out a handle the
 *                      peers resolve to the old, freed memory (measured: wrong sums; tools/mesh_stress.py --empty-cache).              */
#define RSX_MESH_DESC_BYTES 512
typedef struct rsx_mesh rsx_mesh;
int rsx_mesh_alloc(int64_t bytes, void **out);
int rsx_mesh_free(void *p);
int rsx_mesh_alloc_refused(void);                                        /* allocations set aside because they could not be exported */
int rsx_mesh_local(float *Q, float *G, int64_t rows, int d, void *desc_out, rsx_mesh **out);
int rsx_mesh_connect(rsx_mesh *m, int rank, int world, const void *all_desc);
int rsx_mesh_exchange_apply(rsx_mesh *m, int64_t first_row, int64_t rows, float lr, rsx_stream_t stream);
int rsx_mesh_set_wait_limit(rsx_mesh *m, double seconds);
int rsx_mesh_info(const rsx_mesh *m, int *rank, int *world, int64_t *exchanges);
int rsx_mesh_export_retries(const rsx_mesh *m);                        /* -1 for NULL */
int rsx_mesh_check(rsx_mesh *m, rsx_stream_t stream);
void rsx_mesh_destroy(rsx_mesh *m);

/* ---- the native batch loop -----------------------------------------------------------
 * Replaces the reference's inner training loop and the generator feeding it:
 *   models/MF.py:61-72         for b, (users, pos, neg) in enumerate(batch_generator):
 *                                  zero_grad / process_one_batch / backward / optimizer.step
 *                                  epoch_loss += batch_loss
 *   data/generators.py:206-224 PairwiseGenerator.__iter__ (permutation, slicing, H2D copies)
 * rsx_bpr_trainer_run queues n_steps steps on `stream` without returning to the caller's
 * interpreter in between.  Per step: the sampler of step t+1 runs on the trainer's own side
 * stream beside the step kernel of step t; then [exchange]; then rsx_apply_item_grad.  Every
 * kernel is the one behind the stand-alone entry points above, with the same arguments a caller
 * driving rsx_bpr_sample / rsx_bpr_step / rsx_apply_item_grad by hand would pass (step index t,
 * the running position in the user permutation, neg_key derived from (seed_key, t)), so n native
 * steps equal n hand-driven steps.  All device buffers are BORROWED (caller keeps them alive until
 * the stream has drained and the trainer is destroyed); the trainer owns host state, one side
 * stream and a few events, bound to the device current at creation.
 *
 * config
 *   P [num_users x d] (this rank's user rows), Q, G [num_items x d] (G zero before the first step)
 *   indptr / indices     CSR of this rank's users (see rsx_bpr_sample)
 *   batch                largest batch a run may ask for; triplets = int32 [RSX_TRAINER_SLOTS][3][batch] scratch
 *                        (the batch being consumed and the ones sampled ahead)
 *   seed                 sampler seed; seed_key keys the per-step negative-block permutation
 *   neg_block            0 = independent uniform negatives, plain layout.  c > 0 = sorted layout with
 *                        stratified negatives and on-chip summation, engaged for runs whose batch is
 *                        >= 2 * num_items (below that every step of the run takes the plain layout);
 *                        needs sample_ws of rsx_bpr_sample_workspace(batch, num_items) bytes;
 *                        user_sig / item_cdf optional accelerators (built for this CSR and neg_block)
 *   hot_slot / G_hot / hot_items / n_hot / hot_replicas   popular-row replicas or all NULL / 0
 *   loss_acc             nullable float[RSX_LOSS_SLOTS]: sum of softplus(-x) over ALL triplets of ALL
 *                        steps is added (epoch_loss of MF.py:70 = sum(slots) / batch for equal batches)
 *   exchange_begin / exchange_end / exchange_ctx   NULL on one GPU.  With user sharding the caller's
 *                        collective: exchange_begin(ctx) is called when G (folded) is complete on
 *                        `stream` and must start its all-reduce(sum) over the ranks; exchange_end(ctx)
 *                        must make `stream` wait for the reduced G.  Return 0 on success.  Between the
 *                        two the trainer queues the next step's sampler and, if two_pass != 0, the
 *                        user half of the step (RSX_ITEMS_ONLY before, RSX_USERS_ONLY under the exchange).
 *   sort_min_batch       > 0: runs whose batch is at least this large but below 2 * num_items (where neg_block does
 *                        not engage) still order the batch by positive item (RSX_SAMPLE_SORT_POS with independent
 *                        uniform negatives) and step with RSX_BATCH_SORTED; needs sample_ws like neg_block
 *   exchange_applies     != 0: exchange_end also UPDATES Q and leaves G zero (e.g. reduce-scatter of G,
 *                        each rank applying its own shard of item rows, all-gather of the updated rows);
 *                        the trainer then does not call rsx_apply_item_grad itself.
 *   step0 / epoch_pos0   starting step index and position in the user permutation
 *   comm / exchange_kind     the exchange issued BY THE LIBRARY over RCCL on a trainer-owned high-priority stream
 *                        (instead of the callbacks; give one or the other).  RSX_EXCHANGE_ALLREDUCE: in-place
 *                        all-reduce(sum) of G, then every rank applies the identical Q -= lr G.
 *                        RSX_EXCHANGE_SCATTER_GATHER: reduce-scatter of G -> the rank applies ITS shard of item rows
 *                        -> all-gather of the updated Q rows (replicas identical by construction); Q and G must then
 *                        hold world * ceil(num_items / world) rows (item_rows_padded), the tail zero.
 *   chunks / items_real / chunk_pos / progress   chunks > 1: every step as `chunks` independent pipelines over item
 *                        ranges (see "item chunks" above): the tables are in the caller's relabelled item space,
 *                        num_items = chunks * rsx_chunk_rows(items_real, chunks, neg_block); chunk_pos int64
 *                        [RSX_TRAINER_SLOTS][chunks + 1] and progress uint32 [RSX_PROGRESS_WORDS] are device
 *                        scratch.  Range k's kernel, its all-reduce (when comm is given; all ranges' collectives on ONE
 *                        trainer-owned stream, in range order on every rank) and its apply form a chain on the trainer's
 *                        stream for range k; range k of the next step follows its own apply, whatever the other ranges'
 *                        exchanges are doing.  Every batch of such a trainer (<= 2^21 triplets) takes this form; needs
 *                        item_cdf and the sampler workspace; neg_block = 0 selects the form without blocks; not combined with two_pass / stale_exchange / exchange_begin, exchange_end /
 *                        RSX_EXCHANGE_SCATTER_GATHER.
 *   exchange_range       (chunks > 1, no comm) the caller's collective for ONE item range, in place of the library's RCCL
 *                        all-reduce: exchange_range(ctx, k, G_rows, n, stream) is called while the step is being queued, once per
 *                        range and step, in range order (the same order on every rank); it must QUEUE on `stream` -- the
 *                        trainer's collective stream, already ordered behind range k's kernel and the fold of its popular rows
 *                        -- an in-place all-reduce(sum) over the ranks of the n floats at G_rows (= G + k * chunk_rows * d,
 *                        padding rows included), so that work queued on `stream` afterwards (the range's apply) sees the
 *                        reduced rows.  It may block the calling thread (a host-staged collective such as gloo does), never
 *                        the other streams.  Return 0 on success.  This is how the schedule that rsx_comm runs over RCCL is
 *                        driven over any other transport (torch.distributed: tests/test_sharded_gloo.py runs it with two
 *                        ranks).
 *   mesh                 (no comm, no callbacks) the library's own exchange (rsx_mesh_*) over the tables Q and G of THIS config:
 *                        one rsx_mesh_exchange_apply per step, or per item range with chunks > 1 (on the trainer's collective
 *                        stream, under the other ranges' kernels); it applies the summed gradient itself.  Not with two_pass /
 *                        stale_exchange.
 *   csc                  (optional) built by rsx_bpr_build_csc from THIS config's indptr / indices: steps whose batch holds every
 *                        user once (batch == num_users, starting a pass) and take an ordered layout (neg_block engaged, sort_min_batch,
 *                        or chunks > 1) are sampled by rsx_bpr_sample_csc instead of the bucket passes; sample_ws must also hold
 *                        rsx_bpr_sample_csc_workspace(nnz) bytes.  Other steps (short batches) sample as without it.
 *   touched              (optional, unsharded trainers) device bytes [num_items], zero-filled by the caller once.  Where a step runs the
 *                        plain kernel on a batch small against the catalog (2 * batch <= num_items; rsx_set_option "touched_apply"), the
 *                        kernel marks the rows of G it adds to and the apply visits the marked (and the replicated) rows only instead of
 *                        sweeping G -- at 100 000 x 128 the sweep costs 22-28 us whether the batch held 256 triplets (the reference's default
 *                        batch, config.py) or 65 536.  Same arithmetic per element: the tables are bit-identical with and without it.
 *   stale_exchange / G_alt   OPT-IN, needs an exchange (callbacks or comm) and a second zeroed [num_items x d] buffer.
 *                        != 0: the exchange of step t's item gradients travels under the step kernel of
 *                        step t+1, which therefore reads an item table that lacks step t's update (ONE STEP
 *                        STALE: no longer the reference's batch-synchronous step, MF.py:64-68; user rows are
 *                        never stale).  The steps of a trainer alternate between G (first) and G_alt; the
 *                        callbacks come in the order begin(t), end(t-1), begin(t+1), end(t), ...: begin starts
 *                        the collective on the buffer of the step just computed, end finishes the OLDEST one in
 *                        flight.  Every rsx_bpr_trainer_run ends the last exchange and applies it before it
 *                        returns, so nothing is left unapplied between calls.  two_pass is ignored.
 * rsx_bpr_trainer_run(n_steps, batch <= config batch, global_batch = sum of the ranks' batches,
 *   time_every): time_every > 0 brackets the step kernel of every time_every-th step with HIP events
 *   on the stream it is launched on; rsx_bpr_trainer_kernel_ms returns their mean once the stream has drained.  A chunked
 *   step launches one kernel per item range, each on its own stream: every range's kernel gets its own pair of events and
 *   the figure per step is the SUM of the ranges' kernel durations (ranges that overlap on the chip are counted twice:
 *   the figure never flatters the kernel).
 * rsx_bpr_trainer_state: next step index and the permutation position the next batch starts from.
 * rsx_bpr_trainer_seek: set both (drops a batch sampled ahead).
 * rsx_bpr_trainer_last_batch: device pointers of the triplets the most recent step consumed (valid
 *   until the next rsx_bpr_trainer_run) and the neg_block / neg_key it ran with: lets a test
 *   replay exactly what a native step did.                                                   */
#define RSX_TRAINER_SLOTS 3
typedef int (*rsx_exchange_fn)(void *ctx);
typedef int (*rsx_exchange_range_fn)(void *ctx, int range, float *G_rows_dev, int64_t n, rsx_stream_t stream);

typedef struct rsx_bpr_trainer_config {
    float *P;
    float *Q;
    float *G;
    int64_t num_users;
    int64_t num_items;
    int32_t d;
    float lr;
    const int64_t *indptr;
    const int32_t *indices;
    int64_t batch;
    uint64_t seed;
    uint64_t seed_key;
    int32_t neg_block;
    int32_t two_pass;
    void *sample_ws;
    int64_t sample_ws_bytes;
    const uint64_t *user_sig;
    const uint32_t *item_cdf;
    int32_t *triplets;
    const int32_t *hot_slot;
    float *G_hot;
    const int32_t *hot_items;
    int32_t n_hot;
    int32_t hot_replicas;
    float *loss_acc;
    rsx_exchange_fn exchange_begin;
    rsx_exchange_fn exchange_end;
    void *exchange_ctx;
    int32_t exchange_applies;
    int32_t sort_min_batch;
    int64_t step0;
    int64_t epoch_pos0;
    float *G_alt;
    int32_t stale_exchange;
    int32_t exchange_kind;          /* with comm: RSX_EXCHANGE_ALLREDUCE / RSX_EXCHANGE_SCATTER_GATHER       */
    rsx_comm *comm;
    int64_t item_rows_padded;       /* RSX_EXCHANGE_SCATTER_GATHER: rows of Q and G (world * shard); else 0   */
    int32_t chunks;                 /* 0 / 1 = off                                                           */
    int32_t reserved0;
    int64_t items_real;
    int64_t *chunk_pos;
    uint32_t *progress;
    rsx_exchange_range_fn exchange_range;   /* chunks > 1 without comm: the caller's all-reduce of one item range  */
    rsx_mesh *mesh;                         /* the library's own exchange over xGMI (see rsx_mesh_*), or NULL        */
    const rsx_csc *csc;                     /* the CSC walk for whole-pass batches (see rsx_bpr_sample_csc), or NULL  */
    uint8_t *touched;                       /* row marks of the small-batch apply: [num_items] bytes, zero; or NULL   */
} rsx_bpr_trainer_config;

#define RSX_EXCHANGE_ALLREDUCE 1
#define RSX_EXCHANGE_SCATTER_GATHER 2

typedef struct rsx_bpr_trainer rsx_bpr_trainer;

int rsx_bpr_trainer_create(const rsx_bpr_trainer_config *cfg, rsx_bpr_trainer **out);
void rsx_bpr_trainer_destroy(rsx_bpr_trainer *t);
int rsx_bpr_trainer_run(rsx_bpr_trainer *t, int64_t n_steps, int64_t batch, int64_t global_batch,
                        int time_every, rsx_stream_t stream);
int rsx_bpr_trainer_state(const rsx_bpr_trainer *t, int64_t *step, int64_t *epoch_pos);
int rsx_bpr_trainer_seek(rsx_bpr_trainer *t, int64_t step, int64_t epoch_pos, rsx_stream_t stream);
int rsx_bpr_trainer_last_batch(const rsx_bpr_trainer *t, const int32_t **u, const int32_t **i,
                               const int32_t **j, int64_t *batch, int *neg_block, uint64_t *neg_key);
int rsx_bpr_trainer_kernel_ms(const rsx_bpr_trainer *t, double *mean_ms, int64_t *count);
/* waits for `stream`: 0, or RSX_E_INVALID when the chunked steps since the last check saw triplets outside their item range
 * (text via rsx_last_error); reads progress[RSX_PROGRESS_VIOLATIONS] from the device and resets it                  */
int rsx_bpr_trainer_check(rsx_bpr_trainer *t, rsx_stream_t stream);

/* ---- full-catalog scoring + Top-K ----------------------------------------------
 * Replaces
 *   models/MF.py:109-112  predict_batch_users: S = P[users] @ Q.T   (fp32)
 *   models/MF.py:130      pred[eval_pos.nonzero()] = -inf
 *   evaluation/backend/cython/include/func.h:12-31  per-row partial sort, K best
 *                         sorted by descending score (int32 indices)
 * S is computed on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 *
 * rsx_score  : scores_out [num_rows x num_items] = P[user_ids] @ Q^T, then -inf
 *              at the CSR positions of each user's row when mask_indptr != NULL
 *              (CSR is indexed by USER ID, as MF.py:128-130 does).
 * rsx_topk   : per row of `scores` the K largest, descending; ties by lower index.
 *              topk_val_out nullable.  K <= 1024 and K <= num_items.
 * rsx_score_topk : both, tile by tile through `ws`, without exposing the scores.  For catalogs
 *              of >= 32768 items the dense [rows x items] matrix is never formed: the item table
 *              is copied in the order p -> (a p) mod num_items (a coprime to num_items), the first
 *              8192 rows of the copy -- an equidistributed sample of the catalog -- are scored
 *              densely and give each row a lower bound tau of its K-th score plus their own K
 *              candidates, the REST of the copy is scored with an epilogue that keeps only scores
 *              >= tau, and the survivors are masked, sorted and cut to K (every item is scored
 *              exactly once).  Same result as
 *              rsx_score + rsx_topk.  This entry point waits for the stream once before it
 *              returns (it reads a counter of rows that must be re-done densely: massive
 *              exact ties).
 */
int rsx_score(const float *P, const int32_t *user_ids_dev, int64_t num_rows, const float *Q,
              int64_t num_items, int d, const int64_t *mask_indptr_dev,
              const int32_t *mask_indices_dev, float *scores_out, rsx_stream_t stream);

int rsx_topk(const float *scores_dev, int64_t num_rows, int64_t num_items, int K,
             int32_t *topk_idx_out, float *topk_val_out, rsx_stream_t stream);

/* workspace of rsx_score_topk: _d for the row width d it will be called with (the fused path keeps a permuted copy of the item table:
 * num_items * d * 4 bytes of it); the form without d is the bound for d = 256 (at 1M items 512 MB more than d = 128 needs).        */
int64_t rsx_score_topk_workspace(int64_t num_rows, int64_t num_items);
int64_t rsx_score_topk_workspace_d(int64_t num_rows, int64_t num_items, int d);

int rsx_score_topk(const float *P, const int32_t *user_ids_dev, int64_t num_rows, const float *Q,
                   int64_t num_items, int d, const int64_t *mask_indptr_dev,
                   const int32_t *mask_indices_dev, int K, int32_t *topk_idx_out,
                   float *topk_val_out, void *ws, int64_t ws_bytes, rsx_stream_t stream);

/* ---- LightGCN propagation (BASELINE config 5) ------------------
 * Replaces models/LightGCN.py:188-197: all_emb = torch.sparse.mm(A_hat, all_emb), L times,
 * mean over the L+1 layers (:198-200).  A_hat = D^-1/2 [[0,R],[R^T,0]] D^-1/2 is built on the
 * host as the reference builds it (:228-258) and passed as CSR (indptr int64 [N+1], indices
 * int32, vals fp32, N = users + items).  A_hat is symmetric: the backward of the propagation
 * is the same product applied to the gradient.
 * rsx_spmm_plan (HOST): cuts rows into segments of <= max_seg non-zeros (item rows of a
 *   popularity-skewed graph have 10^5+ neighbours); call with NULL outputs for the count.
 * rsx_spmm_csr: Y = A X   (Y is overwritten -- every row must own at least one segment, as rsx_spmm_plan's plans do, or be computed
 *   by rsx_spmm_hot_rows below;
 *   X [N x d] must not alias Y or S_acc);
 *   if S_acc != NULL also S_acc += A X (the running layer sum).
 * rsx_scale: X *= alpha (the 1/(L+1) of the layer mean).                                   */
int64_t rsx_spmm_plan(const int64_t *indptr_host, int64_t num_rows, int max_seg, int32_t *seg_row_out,
                      int64_t *seg_begin_out, int32_t *seg_len_out);
int rsx_spmm_csr(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                 int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                 const float *vals_dev, const float *X, float *Y, float *S_acc, int64_t num_rows, int d,
                 rsx_stream_t stream);
/* rsx_spmm_csr_sparse_rows: the same product for an X most of whose ROWS are entirely zero; x_row_nonzero_dev (uint8 [N]) is
 *   zero for such rows (a non-zero row flagged zero would be dropped: the flags are the caller's claim) and they are not
 *   fetched.  Bit-identical to rsx_spmm_csr.  The first backward product of a LightGCN step: the dense gradient of the
 *   loss has non-zero rows only for the batch's users and items (models/LightGCN.py:83-87).                        */
int rsx_spmm_csr_sparse_rows(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                             int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                             const float *vals_dev, const float *X, const uint8_t *x_row_nonzero_dev, float *Y,
                             float *S_acc, int64_t num_rows, int d, rsx_stream_t stream);
/* rsx_spmm_csr_init: the FIRST product of a propagation (models/LightGCN.py:179-197: embs = [all_emb], then one torch.sparse.mm per
 *   layer; :198-200 their mean), fused with the start of the running layer sum: Y = A X and
 *   S_out = S_init + A X (S_out is overwritten; S_init, usually X itself, is only read) -- instead of copying the source table into
 *   the sum and then adding.  x_row_nonzero_dev: NULL, or the row flags of rsx_spmm_csr_sparse_rows.  Same sums as the two-step form. */
int rsx_spmm_csr_init(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                      int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                      const float *vals_dev, const float *X, const uint8_t *x_row_nonzero_dev, const float *S_init,
                      float *Y, float *S_out, int64_t num_rows, int d, rsx_stream_t stream);
/* rsx_spmm_scale_rows: X[row] *= alpha for the flagged rows (uint8 [num_rows]); alpha = 0 stores zeros.  Two sweeps of a LightGCN
 *   step that concern the batch's rows only (the rows rsx_spmm_mark_batch_rows flagged): the 1 / (L + 1) of the forward layer mean
 *   (the loss reads the propagated tables at those rows and nowhere else), and clearing the dense gradient dL/dOut where it is
 *   non-zero.                                                                                                          */
int rsx_spmm_scale_rows(float *X, const uint8_t *flags_dev, int64_t num_rows, int d, float alpha, rsx_stream_t stream);
/* rsx_spmm_csr_select_rows: the same product for a caller who reads only SOME rows of the result; y_row_wanted_dev (uint8 [N]) is
 *   non-zero for those: the other rows of Y (and of S_acc) are NOT written and hold whatever they held.  The wanted rows are
 *   bit-identical to rsx_spmm_csr's.  The LAST forward product of a LightGCN training step: the loss indexes the propagated
 *   tables by the batch's users and items only (models/LightGCN.py:117-123), so with 65 536 of 1M users in a batch 93 % of the
 *   user rows of that product are never read.                                                                           */
int rsx_spmm_csr_select_rows(const int32_t *seg_row_dev, const int64_t *seg_begin_dev, const int32_t *seg_len_dev,
                             int64_t num_segs, const int64_t *indptr_dev, const int32_t *indices_dev,
                             const float *vals_dev, const float *X, const uint8_t *y_row_wanted_dev, float *Y,
                             float *S_acc, int64_t num_rows, int d, rsx_stream_t stream);
int rsx_scale(float *X, int64_t n, float alpha, rsx_stream_t stream);
/* rsx_spmm_hot_rows: the LONGEST rows of the product by scatter instead of gather.  A popularity-skewed graph puts half of all non-zeros
 *   into a few hundred rows (the popular items); gathered, they re-read the source table once per row.  The caller takes those rows OUT of
 *   the segment plan it hands to the four products above (a row without a segment is not written by them) and describes them by SOURCE
 *   row.  rsx_spmm_hot_rows then stages every source row that has an entry in a hot row through LDS ONCE, adds a * x into register
 *   accumulators -- each of a workgroup's 16 wavefronts owns num_slots / 16 SLOTS -- and adds the wavefronts' rows to Y:
 *   Y[hot rows] = A[hot rows, :] X, and S_acc[hot rows] = (S_init or S_acc)[hot rows] + that: the same sums as the gather, in another
 *   order.  x_row_nonzero_dev / y_row_wanted_dev / S_init: as in the products above (each NULL or the same array the product call got).
 *   Call it after the product over the reduced plan, on the same stream.
 *   The plan (all arrays on the device; recsys_pytorch_amd/rsx.py: SpmmGraph builds it):
 *     num_slots = rsx_spmm_hot_capacity(d) = 16 384 / d; slot s belongs to wavefront s / (num_slots / 16); hot_rows[s] = the row of Y
 *       behind it, or -1; a very long row may own several slots (its entries dealt round) so that no wavefront carries it alone;
 *       uniq_rows = the distinct hot rows;
 *     src_rows: ascending rows of X with at least one hot entry, cut into CHUNKS of chunk_rows = rsx_spmm_hot_chunk_rows(d) rows;
 *     entries grouped by (chunk, wavefront): cw_ptr[chunk * 16 + w] .. cw_ptr[chunk * 16 + w + 1], and inside such a group SORTED BY
 *       SLOT: rw_off[(chunk * 16 + w) * (R + 1) + r] .. [.. + r + 1] (R = num_slots / 16, relative to the group's first entry) are the
 *       entries of the wavefront's slot r; an entry is  ent_code = (row inside the chunk) | (slot inside the wavefront) << 8  and
 *       ent_val = A[hot row, source row].                                                                                         */
typedef struct rsx_spmm_hot {
    int32_t num_slots;           /* rsx_spmm_hot_capacity(d)                                                        */
    int32_t chunk_rows;          /* rsx_spmm_hot_chunk_rows(d)                                                      */
    int32_t num_uniq;            /* distinct hot rows                                                               */
    int32_t reserved;
    int64_t num_src;             /* source rows with at least one entry in a hot row                                */
    const int32_t *hot_rows;     /* [num_slots]: the row of Y behind every slot, or -1                              */
    const int32_t *uniq_rows;    /* [num_uniq]                                                                      */
    const int32_t *src_rows;     /* [num_src] ascending                                                             */
    const int64_t *cw_ptr;       /* [ceil(num_src / chunk_rows) * 16 + 1]                                           */
    const uint16_t *rw_off;      /* [ceil(num_src / chunk_rows) * 16 * (num_slots / 16 + 1)]                        */
    const uint16_t *ent_code;    /* [entries]                                                                       */
    const float *ent_val;        /* [entries]                                                                       */
} rsx_spmm_hot;
int64_t rsx_spmm_hot_capacity(int d);
int64_t rsx_spmm_hot_chunk_rows(int d);
int rsx_spmm_hot_rows(const rsx_spmm_hot *hot, const float *X, const uint8_t *x_row_nonzero_dev, const uint8_t *y_row_wanted_dev,
                      const float *S_init, float *Y, float *S_acc, int64_t num_rows, int d, rsx_stream_t stream);

/* rsx_spmm_mark_batch_rows: the row flags rsx_spmm_csr_sparse_rows takes, for the gradient of ONE batch of triplets on the
 *   stacked [users; items] table: flags (uint8 [num_rows]) = 0 everywhere, then 1 at u[b], item_offset + i[b], item_offset + j[b]
 *   (triplets with i[b] < 0 are skipped, as the step kernels skip them).                                             */
int rsx_spmm_mark_batch_rows(uint8_t *flags_dev, int64_t num_rows, const int32_t *u_dev, const int32_t *i_dev,
                             const int32_t *j_dev, int64_t batch, int64_t item_offset, rsx_stream_t stream);

/* ---- holdout metrics (HOST function, host pointers) --------------------------------
 * Replaces evaluation/backend/cython/include/holdout.h:20-103 (evaluate_holdout) and its
 * wrapper holdout_func.pyx: Prec@K = hits/K, Recall@K = hits/truth_len,
 * NDCG@K = DCG/iDCG with 1/log2(rank+2).  O(users * max_k * log truth): stays on the CPU
 * as in the reference.  rankings: int32 [users_num x max_k] (rsx_topk output copied to
 * the host); truth as CSR (indptr int64 [users_num+1], indices int32, any order);
 * results: float [users_num x 3*K_len], layout [user][metric*K_len + k] with metrics
 * Prec, Recall, NDCG (holdout.h:72-102).  Users with an empty truth row get NaN for
 * Recall/NDCG exactly like the reference header (0/0).  From 8 192 users on the users are cut
 * into contiguous ranges over up to 64 host threads (rsx_eval_loo likewise): every user's numbers
 * are the ones one thread computes; the call returns when all ranges are done.            */
int rsx_eval_holdout(int64_t users_num, const int32_t *rankings, int max_k, const int32_t *Ks,
                     int K_len, const int64_t *truth_indptr, const int32_t *truth_indices,
                     float *results);

/* ---- leave-one-out metrics (HOST function, host pointers; widening row f6) -----------------
 * Replaces evaluation/backend/cython/include/loo.h:19-85 (evaluate_loo) and its python twin
 * evaluation/backend/python/loo.py:11-32: ONE held-out item per user (the first target, loo.h:31);
 * hit_at = its 1-based position in the user's ranking (max_k + 1 when absent);
 * HR@K = [K >= hit_at], NDCG@K = 1 / log2(hit_at + 1) if K >= hit_at else 0.
 * truth: int32 [users_num]; results: float [users_num x 2*K_len], layout [user][metric*K_len + k], metrics HR, NDCG. */
int rsx_eval_loo(int64_t users_num, const int32_t *rankings, int max_k, const int32_t *Ks, int K_len,
                 const int32_t *truth, float *results);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H */
