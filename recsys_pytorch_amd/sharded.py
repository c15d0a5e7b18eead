"""User-sharded BPR step engine: one process per GPU, items replicated.

Partitioning: user rows are split into contiguous blocks,
rank r owns users [r*ceil(U/W), ...); every rank holds the full item table Q and
a full item-gradient buffer G.  A triplet touches one user row and two item
rows, so user-row traffic is strictly local; the only exchange per step is ONE
all-reduce(sum, fp32) of G over RCCL/xGMI (torch.distributed backend "nccl"),
after which every rank applies the identical Q -= lr*G.  The reference has no
multi-device code at all (main.py:24-27 pins one device); the step semantics
that must be preserved are the reference's batch mean (models/MF.py:105): with
W ranks each contributing B_r triplets, every gradient carries
1 / sum_r(B_r), so W ranks on W batches equal one device on the concatenation.

The arithmetic is behind `kernels` (default: the HIP library through
recsys_pytorch_amd.rsx, which refuses host tensors).  Tests inject a CPU
checker there to exercise THIS file's sharding/collective logic under gloo.
"""
import torch
import torch.distributed as dist


def user_block(num_users, rank, world):
    """contiguous block of user ids owned by `rank`: [begin, end)"""
    per = (num_users + world - 1) // world
    begin = min(rank * per, num_users)
    return begin, min(begin + per, num_users)


def pick_neg_block(num_items, max_block, wave_slots, batch=None, min_block=3, floor=2):
    """item block c of the stratified negatives.  The blocked step kernel runs one wavefront per block over
    ~c * batch / num_items positions.

    * many triplets per item (batch / num_items >= 10): the SMALLEST c >= min_block that still gives a wavefront 20 positions.  Short
      wavefronts retire often, and the sampler of the next step -- which time-shares the CUs with this kernel, whose
      wavefronts hold nearly all VGPRs -- finds room continuously instead of at the end of each round.  Measured at
      I = 100K, B = 2**20: c = 2 -> 341 us per step, 3 -> 345, 6 -> 362 (round 3, profiles/r03_exp_sampler_placement.txt); alternating on
      one box in round 6: c = 2: 307 / 329 / 321, c = 3: 303 / 322 / 322, c = 4: 327 / 336 / 337, c = 5: 341 / 343 / 356; d = 64: 2 / 3 / 4 alike.
    * fewer: a wavefront needs its positions to amortise its start, and the kernel's duration is ceil(waves / resident
      slots) ROUNDS (12 500 wavefronts on 6 144 slots take three rounds for two rounds' worth of work): the c in
      [2, max_block] whose last round is fullest; ties go to the larger block.  (B = 262 144 at I = 100K: c = 6 -> 1.54e9
      triplets/s, 3 -> 1.36e9, 2 -> 1.12e9.)

    min_block: lower bound of the first rule, 3.  That is the QUALITY side of the choice: the negatives of one block all come from the
    ~c batch / num_items consecutive positions of ONE wavefront, i.e. from users who share c or so positive items -- the same expectation
    as independent draws, more variance per step.  On a planted-factor dataset in the headline's proportions (tools/sampler_quality.py,
    profiles/r06_sampler_quality.txt) NDCG@10 at the plateau falls short of independent negatives' by 2.4 / 1.4 / 1.4 / 0.9 / 0.8 / 0.45 %
    at c = 2 / 3 / 4 / 6 / 8 / 16 with a step size of 0.1 per triplet, by 0.7 % (c = 2) at 0.05 and by nothing measurable (16 seeds) at 0.02,
    where that model is best.  c = 3 costs nothing on the clock and closes 40 % of that gap, so it is the floor since round 6 (it was 2);
    the item-range pipelines (set_chunks) were at 3 already: two ranges on one GPU take 358 us per step with c = 3, 432-442 with c = 2,
    396-409 with c = 6 (round 3, block F; round 6: c = 3 / 4 / 5 alike, 340 us).  max_block = 2 still gives 2.

    floor: the caller's lower bound of BOTH rules (hparams['neg_block_min']): larger blocks for a large step size, at ~4 % of the
    step per block size from 4 on."""
    if max_block < 2:
        return max(1, int(max_block))
    floor = max(2, min(int(floor), int(max_block)))
    if batch is not None and batch >= 10 * num_items:
        for c in range(max(min(min_block, max_block), floor), max_block + 1):
            if c * batch >= 20 * num_items:
                return c
    best, best_eff = floor, -1.0
    for c in range(floor, max_block + 1):
        waves = -(-num_items // c)
        eff = waves / (-(-waves // wave_slots) * wave_slots)
        if eff >= best_eff - 1e-9:
            best, best_eff = c, max(eff, best_eff)
    return best


def deal_items_to_ranges(mass, cap, rng, heavy=4096):
    """which item range every item sits in for one relabelling round: `assign[item] in [0, C)`, exactly `cap[k]` items in
    range k, the ranges' sampling masses balanced, and -- what the redraw exists for -- a DIFFERENT membership every round
    for the heavy items too (two items meet as positive / negative only while they share a range):

      1. the 16 C heaviest items, in random order, each to a range drawn uniformly among those with a free seat whose load
         stays within 0.9 of a range's share of the heavy mass (none: the lightest) -- the heads of a skewed catalog are not pinned to "one per
         range";
      2. the other heavy items (up to `heavy` in all) greedily to the currently lightest range, visited in mass order
         shuffled inside windows of 8 C (a strict mass order makes neighbours in rank alternate between the ranges for ever),
         ties between equally light ranges broken at random;
      3. the rest at random.

    A function of the global masses and the seeded rng only: identical on every rank."""
    import numpy as np
    mass = np.asarray(mass, dtype=np.float64)
    cap = np.asarray(cap, dtype=np.int64)
    I, C = mass.shape[0], cap.shape[0]
    order = np.argsort(-mass, kind="stable")
    H = min(I, int(heavy))
    T = min(H, 16 * C)
    assign = np.full(I, -1, dtype=np.int64)
    load, cnt = np.zeros(C), np.zeros(C, dtype=np.int64)
    target = 0.9 * float(mass[order[:H]].sum()) / C      # (of the HEAVY mass: the rest arrives evenly, by seats)
    for it in rng.permutation(order[:T]):
        free = cnt < cap
        ok = np.flatnonzero(free & (load + mass[it] <= target))
        k = int(rng.choice(ok)) if ok.size else int(np.argmin(np.where(free, load, np.inf)))
        assign[it] = k; load[k] += mass[it]; cnt[k] += 1
    mid = order[T:H].copy()
    win = 8 * C
    for a in range(0, mid.size, win):
        rng.shuffle(mid[a:a + win])
    for it in mid:
        lo = np.where(cnt < cap, load, np.inf)
        ties = np.flatnonzero(lo <= lo.min() * (1 + 1e-12))
        k = int(ties[rng.integers(ties.size)]) if ties.size > 1 else int(ties[0])
        assign[it] = k; load[k] += mass[it]; cnt[k] += 1
    seats = np.repeat(np.arange(C), cap - cnt)
    rng.shuffle(seats)
    assign[order[H:]] = seats
    return assign


class BPREngine:
    """Owns the step sequence  sample/replay -> bpr_step -> all-reduce(G) -> apply.

    P_local : [U_local x d] fp32, this rank's user rows (local ids 0..U_local)
    Q       : [I x d] fp32, replicated
    """

    def __init__(self, P_local, Q, lr, kernels=None, group=None, user_begin=0, seed=2020, optimizer="sgd",
                 exchange="allreduce", force_sharded=False, comm=None):
        if kernels is None:
            from . import rsx as kernels   # the HIP path; raises if librsx.so is missing
        self.k = kernels
        self.P, self.Q = P_local, Q
        self.G = torch.zeros_like(Q)
        if exchange == "direct" and hasattr(self.k, "mesh_tensor") and Q.is_cuda:
            # the tables the peers will map live in memory the library has ALREADY exported (include/rsx.h: rsx_mesh_alloc) -- a pooled
            # allocation of torch's allocator was refused by the runtime twice.  Like "scatter_gather": a caller that allocated Q goes on
            # with ENGINE.Q
            Qm = self.k.mesh_tensor(*Q.shape)
            Qm.copy_(Q)
            self.Q, self.G = Qm, self.k.mesh_tensor(*Q.shape)
        self.lr = float(lr)
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None or dist.is_initialized()) else 1
        # force_sharded: take the exchange path with a group of ONE rank too (how the RCCL collectives, their
        # streams and the trainer's callbacks are exercised on a one-GPU box; tests/test_sharded_gloo.py)
        self.sharded = self.world > 1 or (bool(force_sharded) and dist.is_initialized())
        # sharded + unique users: the exchange of G travels under the user pass of a two-pass step.  The split
        # costs 165 us per step at the headline shape (measured), less than 51 MB over xGMI can take at any N
        # 
        self.overlap_exchange = self.sharded and exchange != "direct"      # (the mesh exchanges and applies in one go)
        self.user_begin = int(user_begin)
        self.seed = int(seed)
        self.step_count = 0
        self.epoch_pos = 0          # position in the keyed user permutation (sampler)
        self._ws = None
        self._loss = torch.zeros(self.k.RSX_LOSS_SLOTS, dtype=torch.float32, device=Q.device)
        self._trip = None
        self._count = torch.zeros(1, dtype=torch.int64, device=Q.device) if self.sharded else None
        self.optimizer = optimizer
        self.GP = None              # dense user-gradient buffer: Adam, and the pointwise branch (allocated on first use)
        self._pw_t = 0              # Adam's step count on the pointwise branch
        if optimizer == "adam":      # the reference's shipped optimizer (models/MF.py:30): dense moments
            self.GP = torch.zeros_like(P_local)
            self.mP, self.vP = torch.zeros_like(P_local), torch.zeros_like(P_local)
            self.mQ, self.vQ = torch.zeros_like(Q), torch.zeros_like(Q)
        elif optimizer != "sgd":
            raise ValueError(optimizer)
        # how the ranks exchange the item gradients of a step:
        #   "allreduce"      all_reduce(G), then every rank applies the identical Q -= lr*G
        #   "scatter_gather" reduce_scatter(G) -> each rank applies ITS shard of item rows -> all_gather of
        #                    the updated Q rows.  Every item row is computed by exactly one rank and copied
        #                    to the others: the replicas are identical by construction, whatever order the
        #                    collective sums in, and the apply sweep shrinks to 1/W of the table per rank.
        #   "direct"         the library's own full mesh over xGMI (include/rsx.h: rsx_mesh_*): every rank sums ITS slice of the
        #                    rows by reading the peers' G directly (all links at once), applies it, and copies the other slices'
        #                    updated rows from their owners -- reduce-scatter + all-gather without RCCL, replicas identical by
        #                    construction.  HIP kernels only; per step, or per item range with set_chunks.
        if exchange not in ("allreduce", "scatter_gather", "direct"):
            raise ValueError(exchange)
        self.exchange = exchange if (self.sharded and optimizer == "sgd") else "allreduce"
        self._mesh = None           # (rsx.Mesh, the tables it was built over)
        self._work = None
        # comm: an rsx.Comm (RCCL communicator owned by the library).  The native loop then issues the exchange ITSELF on its own
        # stream (include/rsx.h: rsx_bpr_trainer_config.comm) -- no callback into the interpreter inside a run; without it the
        # exchange is this engine's torch.distributed collective handed in as two callbacks (any backend; the CPU tests' gloo)
        self.comm = comm if self.sharded else None
        # > 1: the native loop runs the step as a pipeline over item ranges (include/rsx.h: "item chunks"; set_chunks)
        self.chunks = 0
        self._relabel = None
        # The range an item sits in decides which negatives its positives are paired with.  One relabelling for a whole fit
        # would mean that a user whose few positives fall into k < C ranges NEVER meets the other ranges' items as negatives;
        # so the relabelling is redrawn (seeded with a round counter: identical on every rank) once a trainer has run
        # `redraw_ranges_every` steps on it -- between native runs (adopt()), never inside one.  0 = keep one relabelling.
        self.redraw_ranges_every = 64
        self._relabel_round = 0
        self._relabel_step0 = 0
        self._hot_args = None
        # OPT-IN (native loop only): the exchange of step t travels under the step kernel of step t+1, which
        # then reads an item table WITHOUT step t's update -- one step stale, not the reference's
        # batch-synchronous step (models/MF.py:64-68); include/rsx.h: stale_exchange.  Reported separately.
        self.stale_exchange = False
        self._G_alt = self._Gp_alt = None
        self._pending = []          # exchanges begun and not ended, oldest first: (work, gradient buffer)
        self._begin_step = 0        # exchanges begun by the current native trainer (picks the buffer by parity)
        # decided ONCE from the backend: gloo (the CPU tests) has no reduce_scatter and takes the same shard out of an
        # all_reduce; on RCCL a failing collective propagates (never silently another collective than the peers issue)
        self._has_reduce_scatter = self.sharded and dist.get_backend(group) != "gloo"
        if self.exchange == "scatter_gather":
            self._setup_item_shards()
        self.hot = None
        self.neg_block = 0          # > 0: negatives stratified by item block, batch sorted by positive item
        self.use_item_cdf = True    # order the batch through the item-CDF buckets (False: device radix sort)
        # Sampled batches of at least this many triplets are ordered by positive item also when the negatives are
        # NOT blocked (neg_block == 0 asked for, or B < 2 I), and the step then sums runs of equal positives in
        # registers (include/rsx.h: RSX_BATCH_SORTED).  Default: from 2 triplets per item on, like neg_block, and
        # in any case from 2^19 triplets on.  Measured on MI355X (d = 128, Zipf): I = 100K, B = 1M, independent
        # negatives 963 -> 650 us per step; configs[3] slice (I = 1M, B = 1.25M) 1454 -> 1183 us.  Small batches
        # do not pay: at B = 65 536 / 262 144 (I = 100K) the step kernel gains nothing (64 vs 70 us, 173 us) while the
        # ordering sampler (built for million-triplet batches: 16 workgroups at 65 536) takes 114 / 222 us and
        # becomes the critical path (step 101 -> 132 us, 190 -> 244 us).  0 = never.
        self.sorted_min_batch = min(2 * Q.shape[0], 1 << 19)
        self._sample_ws = {}        # sampler scratch, one per stream role ("main" / "side"): never shared
        self._csr = None            # the CSR tensors the static sampler tables below were built from
        self._sig = self._cdf = None
        # OPT-IN (round 6): whole-pass batches (batch == this rank's users) of an ordered layout sampled by ONE walk over the transposed
        # interaction matrix instead of the bucket passes (include/rsx.h: rsx_bpr_sample_csc); built on first use, per CSR.  Exact and
        # tested like the bucket passes; measured beside the step kernel it is a wash -- headline 339-344 vs 335-337 us per step, d = 64
        # 213-218 vs 213-215, configs[3] slice 1149-1156 vs 1166-1169 (profiles/r06_exp_csc_sampler.txt) -- so the default stays
        self.use_csc = False
        self._csc = None            # (rsx.Csc, indptr, indices)
        self._bufs = None           # double-buffered triplets for the overlapped sampler
        self._side = None           # ONE side stream for the engine's lifetime

    def set_neg_block(self, batch, max_block=8, min_block=2):
        """enable the on-chip gradient summation (blocked negatives + batch sorted by positive
        item, include/rsx.h: neg_block / RSX_SAMPLE_SORT_POS) when every item row gets >= 2
        updates per step; below that there is nothing to combine.  The block size c in [min_block, max_block]
        is `pick_neg_block`'s choice for this batch size (min_block: the caller's floor -- larger blocks mix the
        negatives of more positive items, see pick_neg_block)."""
        self._nb_args = (int(batch), int(max_block), int(min_block))
        nb = pick_neg_block(self.Q.shape[0], int(max_block), self._wave_slots(), int(batch), 3,
                            floor=int(min_block)) if batch >= 2 * self.Q.shape[0] else 0
        if nb != self.neg_block:
            self._csr = None        # the user signatures depend on neg_block: rebuild on next use
        self.neg_block = nb
        return self.neg_block

    def _wave_slots(self):
        """wavefronts of the blocked step kernel resident at a time: CUs x 4 SIMDs x 6 (csrc/rsx_bpr.hip)"""
        try:
            dev = self.Q.device.index or 0
            return int(self.k.device_info(dev)["compute_units"]) * 24
        except Exception:           # noqa: BLE001 -- a kernels stand-in without device_info (CPU tests)
            return 256 * 24

    def _bind_csr(self, indptr, indices):
        """static per-CSR sampler tables (user signatures, item CDF).  The engine keeps references to
        the very tensors they were built from, so `is` identifies the CSR: an address recycled by the
        caching allocator for ANOTHER CSR can never be mistaken for it (a stale signature would
        accept positives as negatives)."""
        if self._csr is not None and self._csr[0] is indptr and self._csr[1] is indices and self._csr[2] == self.neg_block:
            return
        self._sig = (self.k.build_signature(indptr, indices, self.neg_block)
                     if (self.neg_block and hasattr(self.k, "build_signature")) else None)
        self._cdf = self.k.build_item_cdf(indptr, indices, self.Q.shape[0]) if hasattr(self.k, "build_item_cdf") else None
        self._csr = (indptr, indices, self.neg_block)

    def _csc_for(self, indptr, indices, num_items, batch):
        """the rsx.Csc of this CSR when `batch` is a whole pass over this rank's users and the kernels have the walk, else None"""
        if not (self.use_csc and hasattr(self.k, "Csc")) or batch != indptr.numel() - 1:
            return None
        c = self._csc
        if c is None or c[1] is not indptr or c[2] is not indices or c[0].num_items != num_items:
            self._csc = c = (self.k.Csc(indptr, indices, num_items), indptr, indices)
        return c[0]

    def _csc_kw(self, indptr, indices, num_items, batch):
        c = self._csc_for(indptr, indices, num_items, batch)
        return {"csc": c} if c is not None else {}

    def _eff_neg_block(self, batch):
        """the block size a SAMPLED step of this batch size runs with: blocked negatives engage from two triplets per
        item on, exactly the rule of the native loop (csrc/rsx_train.hip: effective_neg_block), so that an epoch's
        short last batch takes the same layout on both paths.  (sample() / step() with an explicit neg_block are the
        caller's business: the kernels are exact on any triplets.)"""
        return self.neg_block if (self.neg_block and batch >= 2 * self.Q.shape[0]) else 0

    def _neg_key(self, step, nb=None):
        """per-step key of the negative-block permutation (nonzero); 0 = identity when not sorting"""
        if not (self.neg_block if nb is None else nb):
            return 0
        z = (self.seed * 0x9E3779B97F4A7C15 + (step + 1) * 0xD1B54A32D192ED03) & (2**64 - 1)
        z ^= z >> 31
        return z | 1

    def set_hot_items(self, item_counts, num_hot=256, replicas=None):
        """spread the gradients of the `num_hot` most popular items over `replicas` private rows
        (contention relief at the atomic unit, include/rsx.h:rsx_bpr_step hot_slot_dev).  replicas=None: 32 where the blocked
        kernel runs (set_neg_block came first and engaged), 16 otherwise -- measured, us per step with 16 / 32 / 64 replicas:
        headline 338 / 329 / 344, d = 64 239 / 236, B = 65 536 (plain kernel) 101 / 105 / 113
        (profiles/r03_exp_sampler_placement.txt, block J)"""
        if not replicas:
            replicas = 32 if self.neg_block else 16
        self.hot = self.k.HotItems(item_counts, num_hot, replicas, self.Q.shape[1], self.Q.device) if num_hot > 0 else None
        self._hot_args = (torch.as_tensor(item_counts), int(num_hot), int(replicas)) if num_hot > 0 else None
        self._relabel = None

    # -- the step as a pipeline over item ranges (include/rsx.h: "item chunks") ---------------------------------
    def set_chunks(self, chunks):
        """chunks > 1 (native loop, SGD): the item rows are cut into `chunks` ranges and the exchange (when sharded: issued by the
        library through `comm`, or this engine's torch.distributed all-reduce handed in range by range) and the apply of a range
        travel under the other ranges' kernels.  With blocked negatives engaged (batch >= 2 items) a position's negative comes
        from an item block of its positive's range; below that (include/rsx.h "item chunks", neg_block = 0) from all real items of
        the range.  The engine then trains on a RELABELLED item space -- a seeded permutation of the item ids, the same on every
        rank, redrawn every `redraw_ranges_every` steps between native runs -- held in its own tables; `adopt()` /
        `sync_items()` copy the item rows back into Q.  What changes for the model: a user's negative is uniform over the items
        of the range its sampled positive fell in (1/chunks of the catalog, another one after every redraw) instead of over the
        whole catalog."""
        chunks = int(chunks)
        if chunks > 1 and (self.optimizer != "sgd" or self.exchange not in ("allreduce", "direct")):
            raise ValueError("chunks > 1 needs SGD and exchange='allreduce' (sharded: the library's communicator `comm`, or without "
                             "one this engine's torch.distributed all-reduce handed in range by range)")
        if chunks != self.chunks:
            self._relabel = None
        self.chunks = chunks if chunks > 1 else 0
        if self.neg_block and getattr(self, "_nb_args", None):      # the block size depends on it (pick_neg_block: min_block)
            self.set_neg_block(*self._nb_args)
        return self.chunks

    def _build_relabel(self, indptr, indices):
        """tables of the relabelled item space for THIS CSR and neg_block (cached by the identity of the CSR tensors)"""
        r = self._relabel
        if (r is not None and r["csr"][0] is indptr and r["csr"][1] is indices and r["nb"] == self.neg_block and r["C"] == self.chunks
                and r["round"] == self._relabel_round):
            return r
        I, d = self.Q.shape
        C, nb, dev = self.chunks, self.neg_block, self.Q.device
        Ic = self.k.chunk_rows(I, C, nb)
        base, rem = divmod(I, C)
        # Which item goes to which range.  The ranges should carry the same share of the batch (their kernels then take
        # the same time and finish staggered by their priorities alone), so the relabelling balances the SAMPLING MASS of
        # the ranges -- item i is the sampled positive with weight sum over its users of 1 / deg(u), summed over the ranks --
        # not just their sizes (deal_items_to_ranges: the heads at random within the target, the other heavy items greedily to
        # the lightest range in a shuffled order, the rest at random -- every round another membership, for the heavy items
        # too).  Seeded, and a function of the global masses only: identical on every rank.
        import numpy as np
        U = indptr.numel() - 1
        # (summed in FIXED POINT, 2^-30 per unit: integer sums do not depend on the order the device adds them in.  As float64 sums
        #  the masses of two launches differed in their last bits -- torch's index_add_ is atomic -- and items of equal count, which
        #  tie exactly, were dealt differently from launch to launch: a seeded run was not reproducible.  Found by the round-5
        #  random-shape tests at I = 782; the integer all-reduce is exact too, whatever order a backend reduces in)
        deg = (indptr[1:] - indptr[:-1]).double()
        w = torch.repeat_interleave(torch.round((1.0 / deg.clamp_min(1.0)) * float(1 << 30)).to(torch.int64), indptr[1:] - indptr[:-1])
        mass = torch.zeros(I, dtype=torch.int64, device=dev).index_add_(0, indices.long(), w)
        if self.sharded and self.world > 1:
            m = mass.cpu() if dist.get_backend(self.group) == "gloo" else mass
            dist.all_reduce(m, group=self.group)
            mass = m.to(dev)
        mass = mass.cpu().numpy().astype(np.float64) / float(1 << 30)
        rng = np.random.default_rng(self.seed * 7919 + 13 + 104729 * self._relabel_round)
        cap = np.array([base + (k < rem) for k in range(C)], dtype=np.int64)      # real items per range
        assign = deal_items_to_ranges(mass, cap, rng)
        perm_parts = []
        for k in range(C):
            members = np.flatnonzero(assign == k)
            rng.shuffle(members)
            perm_parts.append(members)
        perm = torch.from_numpy(np.concatenate(perm_parts))                        # the items in rank order
        counts = torch.from_numpy(cap)
        starts = torch.cumsum(counts, 0) - counts
        which = torch.repeat_interleave(torch.arange(C), counts)
        rank_of_pos = which * Ic + (torch.arange(I) - starts[which])
        item_rank = torch.empty(I, dtype=torch.int64)
        item_rank[perm] = rank_of_pos
        rank_item = torch.full((C * Ic,), -1, dtype=torch.int64)
        rank_item[rank_of_pos] = perm
        item_rank, rank_item = item_rank.to(dev), rank_item.to(dev)
        # the CSR with relabelled columns, rows sorted again
        rows = torch.repeat_interleave(torch.arange(U, device=dev), indptr[1:] - indptr[:-1])
        key = rows * (C * Ic) + item_rank[indices.long()]
        key = torch.sort(key).values
        indices_m = (key - rows * (C * Ic)).to(torch.int32).contiguous()
        del rows, key
        real = rank_item >= 0
        Qm = torch.zeros(C * Ic, d, dtype=self.Q.dtype, device=dev)
        hot = None
        if self._hot_args is not None:
            cnt, num_hot, replicas = self._hot_args
            cm = torch.zeros(C * Ic, dtype=cnt.dtype)
            cm[rank_of_pos] = cnt.cpu()[perm]
            hot = self.k.HotItems(cm, num_hot, replicas, d, dev)
        direct = self.exchange == "direct" and hasattr(self.k, "mesh_tensor") and Qm.is_cuda
        if direct:                                       # (the relabelled tables are what the peers map: memory from rsx_mesh_alloc)
            self.close_mesh()                            # (collective: the mesh over the last round's tables, before they are dropped)
            Qm = self.k.mesh_tensor(C * Ic, d)
        r = {"csr": (indptr, indices), "nb": nb, "C": C, "Ic": Ic, "round": self._relabel_round, "item_rank": item_rank, "rank_item": rank_item, "real": real,
             "indices": indices_m, "Q": Qm, "G": self.k.mesh_tensor(C * Ic, d) if direct else torch.zeros_like(Qm), "hot": hot,
             "sig": self.k.build_signature(indptr, indices_m, nb) if nb else None,
             "cdf": self.k.build_item_cdf(indptr, indices_m, C * Ic)}
        self._relabel = r
        return r

    def _items_to_relabelled(self):
        r = self._relabel
        r["Q"][r["real"]] = self.Q[r["rank_item"][r["real"]]]

    def sync_items(self):
        """copy the item rows of the relabelled table (a chunked native run trains there) back into Q"""
        r = self._relabel
        if r is not None and self.chunks:
            self.Q[r["rank_item"][r["real"]]] = r["Q"][r["real"]]

    def relabel_due(self):
        """a chunked trainer has run its share of steps on the current relabelling: the caller should close it and take a new
        one from native_trainer(), which then draws the next relabelling (MF.fit does, once per epoch at most)"""
        return bool(self.chunks and self._relabel is not None and self._relabel["round"] != self._relabel_round)

    def _exchange_range(self, k, first_row, rows, stream):
        """include/rsx.h: exchange_range -- the all-reduce of ONE item range's gradient rows, queued on the trainer's
        collective stream (a raw HIP stream handle; 0 for a CPU stand-in): the step of the chunked native loop that the
        library issues over RCCL itself when it has a communicator, here over torch.distributed (any backend)"""
        G = self._relabel["G"][first_row:first_row + rows]
        if G.is_cuda:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=G.device)):
                dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group)

    # -- the exchange of a step's item gradients ------------------------------------------------
    def _setup_item_shards(self):
        """equal item shards for reduce_scatter / all_gather: the tables are re-homed into buffers padded
        to W * ceil(I / W) rows (self.Q / self.G stay [I x d] views of them; a caller that allocated Q
        must go on using ENGINE.Q, which is the same storage only when I divides evenly)"""
        W, r = self.world, dist.get_rank(self.group)
        I, d = self.Q.shape
        self._shard = (I + W - 1) // W
        rows = self._shard * W
        if rows != I:
            Qp = torch.zeros(rows, d, dtype=self.Q.dtype, device=self.Q.device)
            Qp[:I] = self.Q
            self.Q = Qp[:I]
        else:
            Qp = self.Q
        self._Qp = Qp
        self._Gp = torch.zeros(rows, d, dtype=self.Q.dtype, device=self.Q.device)
        self.G = self._Gp[:I]
        self._mine = slice(r * self._shard, (r + 1) * self._shard)

    def _stale_buffers(self):
        """the second gradient buffer of the one-step-stale exchange (same shape and padding as the first)"""
        if self._G_alt is None:
            if self.exchange == "scatter_gather":
                self._Gp_alt = torch.zeros_like(self._Gp)
                self._G_alt = self._Gp_alt[:self.Q.shape[0]]
            else:
                self._G_alt = torch.zeros_like(self.G)
        return self._G_alt

    def _mesh_over(self, Q, G):
        """the rsx.Mesh over these two tables (collective: every rank builds it at the same point); one at a time -- a mesh over
        other tables (the relabelled item space of another round) replaces it"""
        m = self._mesh
        if m is not None and m[1] is Q and m[2] is G:
            return m[0]
        self.close_mesh()
        if not hasattr(self.k, "Mesh"):
            raise ValueError("exchange='direct' needs the HIP library (rsx.Mesh)")
        mesh = self.k.Mesh(Q, G, group=self.group)
        import os
        if os.environ.get("RSX_MESH_WAIT_S"):       # how long a kernel waits for a peer's signal before it gives up (default 20 s)
            mesh.set_wait_limit(float(os.environ["RSX_MESH_WAIT_S"]))
        self._mesh = (mesh, Q, G)
        return mesh

    def close_mesh(self):
        """collective: checks and releases the mesh (a barrier first -- no peer may still read this rank's tables)"""
        if self._mesh is not None:
            mesh, self._mesh = self._mesh[0], None
            try:
                mesh.check_all()                    # (every rank learns every rank's result before anyone raises)
            finally:
                mesh.close()

    def _exchange_begin(self):
        """the gradient buffer of this step (folded) is complete on the current stream: start the collective.
        Begun exchanges queue up (one deep normally, two with stale_exchange) and end oldest first."""
        alt = self.stale_exchange and (self._begin_step & 1)
        self._begin_step += 1
        if self.exchange == "direct":       # sums, applies and redistributes in one go (Q updated, G zero afterwards)
            self._mesh_over(self.Q, self.G).exchange_apply(0, self.Q.shape[0], self.lr)
            self._pending.append((None, None))
            return
        if self.exchange == "allreduce":
            G = self._G_alt if alt else self.G
            self._pending.append((dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group, async_op=True), G))
            return
        Gp = self._Gp_alt if alt else self._Gp
        if self._has_reduce_scatter:    # in place: this rank's shard of G receives the sum over the ranks
            work = dist.reduce_scatter_tensor(Gp[self._mine], Gp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:                           # gloo (CPU tests): the same shard via all_reduce
            work = dist.all_reduce(Gp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, Gp))

    def _exchange_end(self):
        """the current stream waits for the OLDEST exchange in flight; with "scatter_gather" this also applies the
        own shard and gathers the updated item rows, so no apply sweep follows"""
        work, Gp = self._pending.pop(0)
        if self.exchange == "direct":
            return
        work.wait()
        if self.exchange == "allreduce":
            return
        m = self._mine
        self.k.apply_item_grad(self._Qp[m], Gp[m], self.lr)                # Q -= lr*G on the own rows; zeroes them in G
        Gp[:m.start].zero_()                                               # the other shards hold this rank's partial sums
        Gp[m.stop:].zero_()
        dist.all_gather_into_tensor(self._Qp, self._Qp[m], group=self.group)   # in place: every rank's updated rows

    # -- helpers ---------------------------------------------------------------
    def _workspace(self, batch):
        need = self.k.bpr_step_workspace(self.P.shape[0], batch, self.P.shape[1])
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.P.device)
        return self._ws

    def _global_batch(self, local_batch, global_batch):
        if global_batch is not None:
            return int(global_batch)
        if not self.sharded:
            return int(local_batch)
        self._count.fill_(int(local_batch))
        dist.all_reduce(self._count, group=self.group)
        return int(self._count.item())

    # -- one step on explicit triplets (local user ids) ---------------------------
    def _sorts(self, batch, nb=None):
        """is a sampled batch of this size ordered by positive item (blocked negatives or not)?"""
        return bool(self.neg_block if nb is None else nb) or bool(self.sorted_min_batch and batch >= self.sorted_min_batch
                                                                  and hasattr(self.k, "build_item_cdf"))

    def step(self, u_local, i, j, global_batch=None, users_unique=False, want_loss=True, neg_block=0,
             neg_key=0, batch_sorted=False):
        """returns the device tensor of loss slots (sum_b softplus(-x_b) striped) or None"""
        if self.stale_exchange and self.sharded:
            raise ValueError("stale_exchange is a schedule of the native loop (native_trainer); step() is synchronous")
        B = int(u_local.numel())
        gb = self._global_batch(B, global_batch)
        loss = None
        if want_loss:
            loss = self._loss
            loss.zero_()
        if self.optimizer == "adam":
            if B > 0:
                self.k.bpr_grad(self.P, self.Q, self.GP, self.G, u_local, i, j, 1.0 / gb, loss_acc=loss)
            if self.sharded:
                dist.all_reduce(self.G, op=dist.ReduceOp.SUM, group=self.group)
                if want_loss:
                    dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
            self.step_count += 1
            self.k.adam_apply(self.Q, self.mQ, self.vQ, self.G, self.lr, self.step_count)
            self.k.adam_apply(self.P, self.mP, self.vP, self.GP, self.lr, self.step_count)
            return loss
        kw = {"hot": self.hot} if self.hot is not None else {}
        if neg_block:
            kw["neg_block"], kw["neg_key"] = neg_block, neg_key
        elif batch_sorted and users_unique:
            kw["batch_sorted"] = True
        if self.sharded and users_unique and self.overlap_exchange:
            # two passes over the same triplets (include/rsx.h: RSX_ITEMS_ONLY / RSX_USERS_ONLY): the
            # item pass completes G, whose all-reduce then runs while the user pass updates P --
            # neither pass changes what the other reads, so the step is the same as in one launch
            if B > 0:
                self.k.bpr_step(self.P, self.Q, self.G, u_local, i, j, self.lr, 1.0 / gb, loss_acc=loss,
                                users_unique=True, only="items", **kw)
                if "hot" in kw:
                    self.k.fold_hot_grad(self.G, self.hot)
            self._exchange_begin()
            if B > 0:
                self.k.bpr_step(self.P, self.Q, self.G, u_local, i, j, self.lr, 1.0 / gb,
                                users_unique=True, only="users", **kw)
            self._exchange_end()
            if want_loss:
                dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
            if self.exchange == "allreduce":
                self.k.apply_item_grad(self.Q, self.G, self.lr)
            self.step_count += 1
            return loss
        if B > 0:
            self.k.bpr_step(self.P, self.Q, self.G, u_local, i, j, self.lr, 1.0 / gb, loss_acc=loss,
                            users_unique=users_unique, ws=None if users_unique else self._workspace(B), **kw)
            if "hot" in kw and self.sharded:
                self.k.fold_hot_grad(self.G, self.hot)      # the all-reduce needs the folded G
        if self.sharded:
            # the one exchange of the step: item gradients, summed over ranks (RCCL over xGMI)
            self._exchange_begin()
            self._exchange_end()
            if want_loss:
                dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        if self.hot is not None and not self.sharded:
            self.k.apply_item_grad(self.Q, self.G, self.lr, hot=self.hot)   # replicas folded in the sweep
        elif self.exchange == "allreduce":
            self.k.apply_item_grad(self.Q, self.G, self.lr)
        self.step_count += 1
        return loss

    # -- one step of the POINTWISE branch (models/MF.py:99-102, hparams['pointwise']) -----------
    def pointwise_step(self, u, i, y, loss_func="ce", count_step=True):
        """(user, item, rating) batch, users and items may repeat: dense gradients (rsx_pointwise_grad) + the as-shipped
        Adam or the SGD sweep over both tables.  One GPU (the pointwise branch is not user-sharded)."""
        if self.sharded:
            raise ValueError("the pointwise branch runs on one GPU")
        if self.GP is None:
            self.GP = torch.zeros_like(self.P)
        loss = self._loss
        loss.zero_()
        n = int(u.numel())
        if n > 0:
            self.k.pointwise_grad(self.P, self.Q, self.GP, self.G, u, i, y, 1.0 / n, loss_func=loss_func, loss_acc=loss)
        if count_step:
            self.step_count += 1
        self._pw_t += 1
        if self.optimizer == "adam":
            self.k.adam_apply(self.Q, self.mQ, self.vQ, self.G, self.lr, self._pw_t)
            self.k.adam_apply(self.P, self.mP, self.vP, self.GP, self.lr, self._pw_t)
        else:
            self.k.apply_item_grad(self.Q, self.G, self.lr)      # table-agnostic sweep: theta -= lr * grad; grad = 0
            self.k.apply_item_grad(self.P, self.GP, self.lr)
        return loss

    # -- one step on triplets sampled on the device from this rank's CSR rows -----------
    def _launch_sample(self, indptr, indices, batch, out, step, role="main", nb=None):
        """device sampler (include/rsx.h:rsx_bpr_sample) on the CURRENT stream; users unique
        inside the batch.  A batch never straddles two passes over the user permutation: when
        fewer than `batch` users remain in the pass, the pass restarts (tail dropped).
        `role` names the stream the call is queued on; each role has its own scratch."""
        U = indptr.numel() - 1
        if (self.epoch_pos % U) + batch > U:
            self.epoch_pos = (self.epoch_pos // U + 1) * U
        u, i, j = out
        kw = {}
        nb = self.neg_block if nb is None else nb
        key = self._neg_key(step, nb)
        if self._sorts(batch, nb):
            need = self.k.bpr_sample_workspace(batch, self.Q.shape[0])
            ws = self._sample_ws.get(role)
            if ws is None or ws.numel() < need:
                ws = self._sample_ws[role] = torch.empty(need, dtype=torch.uint8, device=self.Q.device)
            self._bind_csr(indptr, indices)
            kw = {"neg_block": nb, "neg_key": key, "sort_pos": True, "ws": ws}
            if self._sig is not None and nb:
                kw["user_sig"] = self._sig
            if self._cdf is not None and self.use_item_cdf:
                kw["item_cdf"] = self._cdf
        csc = self._csc_for(indptr, indices, self.Q.shape[0], batch) if (kw and self.epoch_pos % U == 0) else None
        if csc is not None:          # the same rule as the native loop (csrc/rsx_train.hip: launch_sample)
            if kw["ws"].numel() < csc.sample_ws_bytes:
                kw["ws"] = self._sample_ws[role] = torch.empty(csc.sample_ws_bytes, dtype=torch.uint8, device=self.Q.device)
            self.k.bpr_sample_csc(csc, indptr, indices, self.Q.shape[0], self.seed + 7919 * self.user_begin, step, u, i, j,
                                  neg_block=nb, neg_key=key, ws=kw["ws"], user_sig=kw.get("user_sig"))
        else:
            self.k.bpr_sample(indptr, indices, self.Q.shape[0], batch, self.seed + 7919 * self.user_begin,
                              step, self.epoch_pos, u, i, j, **kw)
        self.epoch_pos += batch
        return key

    def _triplet_buffers(self, batch):
        mk = lambda: torch.empty(batch, dtype=torch.int32, device=self.Q.device)
        return (mk(), mk(), mk())

    def sample(self, indptr, indices, batch):
        batch = min(int(batch), indptr.numel() - 1)
        if self._trip is None or self._trip[0].numel() != batch:
            self._trip = self._triplet_buffers(batch)
        self.last_neg_key = self._launch_sample(indptr, indices, batch, self._trip, self.step_count)
        return self._trip

    def sampled_step(self, indptr, indices, batch, global_batch=None, want_loss=True):
        batch = min(int(batch), indptr.numel() - 1)
        nb = self._eff_neg_block(batch)
        if self._trip is None or self._trip[0].numel() != batch:
            self._trip = self._triplet_buffers(batch)
        u, i, j = self._trip
        key = self.last_neg_key = self._launch_sample(indptr, indices, batch, self._trip, self.step_count, nb=nb)
        return self.step(u, i, j, global_batch=global_batch, users_unique=True, want_loss=want_loss,
                         neg_block=nb, neg_key=key, batch_sorted=self._sorts(batch, nb))

    def sampled_step_overlapped(self, indptr, indices, batch, global_batch=None, want_loss=True):
        """same result as sampled_step, but the sampler of step t+1 runs on a second HIP stream
        while step t's kernels run (it reads only the CSR, never the tables)."""
        batch = min(int(batch), indptr.numel() - 1)
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.Q.device)
        if self._bufs is None or self._bufs[0]["t"][0].numel() != batch:
            if self._bufs is not None and self._sampled_upto > self.step_count:
                # a batch of the OLD size was sampled ahead (e.g. before an epoch's short last batch):
                # give its positions back to the user permutation, and let everything queued on the
                # side stream finish before its buffers are dropped
                self.epoch_pos = self._bufs[self._cur]["pos_before"]
                main.wait_stream(self._side)
            self._bufs = [{"t": self._triplet_buffers(batch), "ready": None, "free": None, "key": 0, "pos_before": 0}
                          for _ in range(2)]
            self._cur = 0
            self._sampled_upto = self.step_count          # next step index to sample for

        nb = self._eff_neg_block(batch)

        def prefetch(slot):
            buf = self._bufs[slot]
            if buf["free"] is not None:
                self._side.wait_event(buf["free"])        # the step that read this buffer is done
            buf["pos_before"] = self.epoch_pos
            with torch.cuda.stream(self._side):
                buf["key"] = self._launch_sample(indptr, indices, batch, buf["t"], self._sampled_upto, role="side", nb=nb)
                buf["ready"] = torch.cuda.Event()
                buf["ready"].record(self._side)
            self._sampled_upto += 1

        cur = self._cur
        if self._sampled_upto == self.step_count:
            self._side.wait_stream(main)
            prefetch(cur)
        buf = self._bufs[cur]
        main.wait_event(buf["ready"])
        prefetch(cur ^ 1)                                  # sampler of the next step, concurrently
        u, i, j = buf["t"]
        loss = self.step(u, i, j, global_batch=global_batch, users_unique=True, want_loss=want_loss,
                         neg_block=nb, neg_key=buf["key"], batch_sorted=self._sorts(batch, nb))
        buf["free"] = torch.cuda.Event()
        buf["free"].record(main)
        self._cur = cur ^ 1
        return loss

    # -- the native batch loop (include/rsx.h: rsx_bpr_trainer_*) ---------------------------------
    def native_trainer(self, indptr, indices, batch, loss_acc=None):
        """C++ loop over this engine's tables and sampler state: `trainer.run(n, batch)` equals n calls of
        sampled_step_overlapped(batch) (same triplets: same seed, step indices, permutation positions,
        per-step keys, and the same layout rule for a short batch: _eff_neg_block) without returning to Python
        between kernels.  When sharded, the exchange is
        this engine's all-reduce(G) handed in as a pair of callbacks.  SGD only."""
        if self.optimizer != "sgd":
            raise ValueError("the native loop runs the SGD step; optimizer='adam' steps through BPREngine.step")
        batch = min(int(batch), indptr.numel() - 1)
        native = self.sharded and self.comm is not None and self.exchange != "direct"      # the library issues the exchange itself (RCCL)
        kind = {"allreduce": 1, "scatter_gather": 2}[self.exchange] if native else 0      # RSX_EXCHANGE_*
        stale = bool(self.stale_exchange) and self.sharded
        blocked = bool(self.neg_block) and batch >= 2 * self.Q.shape[0]
        # (below two triplets per item the ranges run WITHOUT blocks -- include/rsx.h "item chunks", neg_block = 0: negatives
        #  uniform over the real items of the positive's range -- wherever the ordered layout engages at all)
        unblocked = not self.neg_block and bool(self.sorted_min_batch) and batch >= self.sorted_min_batch
        if self.chunks and (blocked or unblocked) and batch <= (1 << 21) and not stale:
            # the step as a pipeline over item ranges, in the relabelled item space (set_chunks)
            r = self._build_relabel(indptr, indices)
            self._items_to_relabelled()
            self._relabel_step0 = self.step_count
            direct = self.sharded and self.exchange == "direct"
            by_range = {"exchange_range": self._exchange_range} if (self.sharded and not native and not direct) else {}
            if direct:       # the library's own mesh over the relabelled tables (collective; replaces the mesh of the round before)
                by_range = {"mesh": self._mesh_over(r["Q"], r["G"])}
            return self.k.BPRTrainer(self.P, r["Q"], r["G"], indptr, r["indices"], self.lr, batch,
                                     seed=self.seed + 7919 * self.user_begin, seed_key=self.seed, neg_block=self.neg_block,
                                     hot=r["hot"], user_sig=r["sig"], item_cdf=r["cdf"], loss_acc=loss_acc,
                                     comm=self.comm if native else None, exchange_kind=kind, chunks=self.chunks,
                                     items_real=self.Q.shape[0], step0=self.step_count, epoch_pos0=self.epoch_pos,
                                     **self._csc_kw(indptr, r["indices"], r["Q"].shape[0], batch), **by_range)
        sort_min = int(self.sorted_min_batch) if (self.sorted_min_batch and not self.neg_block) else 0
        if self.neg_block or sort_min:
            self._bind_csr(indptr, indices)
        direct = self.sharded and self.exchange == "direct"
        exchange = (self._exchange_begin, self._exchange_end) if (self.sharded and not native and not direct) else None
        self._pending.clear()
        self._begin_step = 0                        # the trainer alternates G / G_alt, G first
        sg = self.exchange == "scatter_gather"
        extra = {}
        if direct:
            extra = {"mesh": self._mesh_over(self.Q, self.G)}
        elif native:
            extra = {"comm": self.comm, "exchange_kind": kind, "item_rows_padded": self._Qp.shape[0] if sg else 0,
                     "num_items": self.Q.shape[0]}
        return self.k.BPRTrainer(self.P, self._Qp if (native and sg) else self.Q,
                                 self._Gp if (native and sg) else self.G, indptr, indices, self.lr, batch,
                                 seed=self.seed + 7919 * self.user_begin, seed_key=self.seed, neg_block=self.neg_block,
                                 hot=self.hot, user_sig=self._sig if self.neg_block else None,
                                 item_cdf=self._cdf if ((self.neg_block or sort_min) and self.use_item_cdf) else None,
                                 sort_min_batch=sort_min,
                                 loss_acc=loss_acc, exchange=exchange, two_pass=self.overlap_exchange and not direct,
                                 exchange_applies=self.exchange == "scatter_gather",
                                 step0=self.step_count, epoch_pos0=self.epoch_pos, **extra,
                                 **(self._csc_kw(indptr, indices, self.Q.shape[0], batch) if (self.neg_block or sort_min) else {}),
                                 **({"G_alt": self._stale_buffers()} if stale else {}))

    def adopt(self, trainer):
        """take over the step counter and permutation position a native run has reached (and, after a chunked run, the
        item rows it trained in the relabelled space)"""
        self.step_count, self.epoch_pos = trainer.state()
        if self._mesh is not None:
            self._mesh[0].check()                   # (synchronises) no wait for a peer's signal gave up
        if getattr(trainer, "chunks", 0) > 1:
            trainer.check()
            self.sync_items()
            if self.redraw_ranges_every and self.step_count - self._relabel_step0 >= self.redraw_ranges_every \
                    and self._relabel is not None and self._relabel["round"] == self._relabel_round:
                self._relabel_round += 1            # the next native_trainer() draws the next relabelling (relabel_due)

    # -- replay of GLOBAL-id triplets: each rank keeps the triplets of its own users -----
    def route(self, u_global, i, j):
        lo, hi = self.user_begin, self.user_begin + self.P.shape[0]
        keep = (u_global >= lo) & (u_global < hi)
        return ((u_global[keep] - lo).to(torch.int32).contiguous(), i[keep].to(torch.int32).contiguous(),
                j[keep].to(torch.int32).contiguous())

    def item_checksum(self):
        """sum of Q in fp64: identical on every rank when the replicas agree"""
        return float(self.Q.double().sum().item())
