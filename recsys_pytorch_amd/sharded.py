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


class BPREngine:
    """Owns the step sequence  sample/replay -> bpr_step -> all-reduce(G) -> apply.

    P_local : [U_local x d] fp32, this rank's user rows (local ids 0..U_local)
    Q       : [I x d] fp32, replicated
    """

    def __init__(self, P_local, Q, lr, kernels=None, group=None, user_begin=0, seed=2020):
        if kernels is None:
            from . import rsx as kernels   # the HIP path; raises if librsx.so is missing
        self.k = kernels
        self.P, self.Q = P_local, Q
        self.G = torch.zeros_like(Q)
        self.lr = float(lr)
        self.group = group
        self.world = dist.get_world_size(group) if (group is not None or dist.is_initialized()) else 1
        self.sharded = self.world > 1
        self.user_begin = int(user_begin)
        self.seed = int(seed)
        self.step_count = 0
        self.epoch_pos = 0          # position in the keyed user permutation (sampler)
        self._ws = None
        self._loss = torch.zeros(self.k.RSX_LOSS_SLOTS, dtype=torch.float32, device=Q.device)
        self._trip = None
        self._count = torch.zeros(1, dtype=torch.int64, device=Q.device) if self.sharded else None

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
    def step(self, u_local, i, j, global_batch=None, users_unique=False, want_loss=True):
        """returns the device tensor of loss slots (sum_b softplus(-x_b) striped) or None"""
        B = int(u_local.numel())
        gb = self._global_batch(B, global_batch)
        loss = None
        if want_loss:
            loss = self._loss
            loss.zero_()
        if B > 0:
            self.k.bpr_step(self.P, self.Q, self.G, u_local, i, j, self.lr, 1.0 / gb, loss_acc=loss,
                            users_unique=users_unique, ws=None if users_unique else self._workspace(B))
        if self.sharded:
            # the one exchange of the step: item gradients, summed over ranks (RCCL over xGMI)
            dist.all_reduce(self.G, op=dist.ReduceOp.SUM, group=self.group)
            if want_loss:
                dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        self.k.apply_item_grad(self.Q, self.G, self.lr)
        self.step_count += 1
        return loss

    # -- one step on triplets sampled on the device from this rank's CSR rows -----------
    def sample(self, indptr, indices, batch):
        """device sampler (include/rsx.h:rsx_bpr_sample); users unique inside the batch.
        A batch never straddles two passes over the user permutation: when fewer than
        `batch` users remain in the pass, the pass restarts (tail dropped)."""
        U = indptr.numel() - 1
        batch = min(int(batch), U)
        if self._trip is None or self._trip[0].numel() != batch:
            mk = lambda: torch.empty(batch, dtype=torch.int32, device=self.Q.device)
            self._trip = (mk(), mk(), mk())
        if (self.epoch_pos % U) + batch > U:
            self.epoch_pos = (self.epoch_pos // U + 1) * U
        u, i, j = self._trip
        self.k.bpr_sample(indptr, indices, self.Q.shape[0], batch, self.seed + 7919 * self.user_begin,
                          self.step_count, self.epoch_pos, u, i, j)
        self.epoch_pos += batch
        return u, i, j

    def sampled_step(self, indptr, indices, batch, global_batch=None, want_loss=True):
        u, i, j = self.sample(indptr, indices, batch)
        return self.step(u, i, j, global_batch=global_batch, users_unique=True, want_loss=want_loss)

    # -- replay of GLOBAL-id triplets: each rank keeps the triplets of its own users -----
    def route(self, u_global, i, j):
        lo, hi = self.user_begin, self.user_begin + self.P.shape[0]
        keep = (u_global >= lo) & (u_global < hi)
        return ((u_global[keep] - lo).to(torch.int32).contiguous(), i[keep].to(torch.int32).contiguous(),
                j[keep].to(torch.int32).contiguous())

    def item_checksum(self):
        """sum of Q in fp64: identical on every rank when the replicas agree"""
        return float(self.Q.double().sum().item())
