"""TEST INFRASTRUCTURE: one BPR step verified in fp64 on the device, at any size.

The closed form of SURVEY section 8 row a6 (models/MF.py:64-68 with SGD), everything taken from the PRE-step
tables:   x_b = <P[u_b], Q[i_b] - Q[j_b]>,  c_b = sigmoid(-x_b) * inv_batch
          P[u_b] += lr * c_b * (Q[i_b] - Q[j_b])           (users unique inside the batch)
          Q      -= lr * G,   G = index_add(i_b, -c_b P[u_b]) + index_add(j_b, +c_b P[u_b])
          loss    = sum_b softplus(-x_b) * inv_batch
`verify_step` measures the error of the UPDATE relative to the largest entry of the update (conftest.delta_err's
definition) -- never against the size of the table, and with no absolute slack: run it with a step size that makes the
update resolvable in fp32 (conftest.resolvable_lr).
"""
import torch

SLICE = 1 << 18          # fp64 temporaries stay at SLICE x d


def fp64_coefficients(P0, Q0, ul, il, jl, inv_batch):
    """x and c = sigmoid(-x) * inv_batch per triplet, fp64, from the pre-step tables (triplets with i < 0 get c = 0)"""
    B = ul.numel()
    x = torch.zeros(B, dtype=torch.float64, device=P0.device)
    live = il >= 0
    for s0 in range(0, B, SLICE):
        sl = slice(s0, min(B, s0 + SLICE))
        i_ = il[sl].clamp_min(0)
        j_ = jl[sl].clamp_min(0)
        x[sl] = (P0[ul[sl]].double() * (Q0[i_].double() - Q0[j_].double())).sum(1)
    c = torch.sigmoid(-x) * float(inv_batch)
    c[~live] = 0.0
    return x, c, live


def verify_step(P0, Q0, P, Q, u, i, j, lr, inv_batch, G=None):
    """returns dict(err_P, err_Q, err_G, loss): update errors relative to the largest update entry.
    P, Q: tables AFTER the step (Q after the apply; pass G -- the gradient buffer BEFORE the apply -- to check it too, then
    Q may be None).  Users must be unique inside the batch."""
    ul, il, jl = u.long(), i.long(), j.long()
    B, d = ul.numel(), P0.shape[1]
    x, c, live = fp64_coefficients(P0, Q0, ul, il, jl, inv_batch)
    loss = float((torch.nn.functional.softplus(-x) * live).sum() * float(inv_batch))
    out = {"loss": loss}
    # item side: dense fp64 index_add
    want_G = torch.zeros(Q0.shape[0], d, dtype=torch.float64, device=P0.device)
    worst_P = biggest_P = 0.0
    for s0 in range(0, B, SLICE):
        sl = slice(s0, min(B, s0 + SLICE))
        keep = live[sl]
        us, is_, js, cs = ul[sl][keep], il[sl][keep], jl[sl][keep], c[sl][keep]
        g = -cs.unsqueeze(1) * P0[us].double()
        want_G.index_add_(0, is_, g)
        want_G.index_add_(0, js, -g)
        dP_want = float(lr) * cs.unsqueeze(1) * (Q0[is_].double() - Q0[js].double())
        dP_got = P[us].double() - P0[us].double()
        if dP_want.numel():
            worst_P = max(worst_P, float((dP_got - dP_want).abs().max()))
            biggest_P = max(biggest_P, float(dP_want.abs().max()))
    out["err_P"] = worst_P / (biggest_P + 1e-300)
    out["max_dP"] = biggest_P
    # rows of users outside the batch (and of skipped triplets) must not move at all
    moved = torch.zeros(P0.shape[0], dtype=torch.bool, device=P0.device)
    moved[ul[live]] = True
    out["untouched_rows_equal"] = bool((P[~moved] == P0[~moved]).all())
    big_G = float(want_G.abs().max())
    if G is not None:
        out["err_G"] = float((G.double() - want_G).abs().max()) / (big_G + 1e-300)
    if Q is not None:
        dQ_want = want_G * (-float(lr))
        worst = 0.0
        for s0 in range(0, Q0.shape[0], SLICE):
            sl = slice(s0, min(Q0.shape[0], s0 + SLICE))
            worst = max(worst, float(((Q[sl].double() - Q0[sl].double()) - dQ_want[sl]).abs().max()))
        out["err_Q"] = worst / (float(lr) * big_G + 1e-300)
        out["max_dQ"] = float(lr) * big_G
    return out
