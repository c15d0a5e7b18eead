"""oracle/gen_golden_pointwise.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

G8: the POINTWISE branch of the reference model (models/MF.py:19-21,48-51,99-102: hparams['pointwise'] = True,
loss_func 'ce' -> F.binary_cross_entropy_with_logits, 'mse' -> F.mse_loss).  Fixtures from the reference's own
process_one_batch + backward + optimizer (as-shipped Adam, and the SGD-swapped optimizer of the north star):
random (user, item, rating) batches with repeated users and items, and batches from the reference's own
PointwiseGenerator on a small interaction matrix (batch_size interactions + ONE sampled negative for EVERY user per
batch, data/generators.py:105-130,79-100).  Asserts oracle == reference while generating.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_pointwise.py
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)
import torch  # noqa: E402
from data.generators import PointwiseGenerator  # noqa: E402  (reference)


def make_ref(U, I, d, P0, Q0, optimizer, lr, loss_func):
    ds = types.SimpleNamespace(num_users=U, num_items=I)
    m = G.MF(ds, {"hidden_dim": d, "pointwise": True, "loss_func": loss_func}, torch.device("cpu"))
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(P0))
        m.item_embedding.weight.copy_(torch.from_numpy(Q0))
    if optimizer == "sgd":  # harness-side swap; reference files untouched
        m.optimizer = torch.optim.SGD(m.parameters(), lr=lr)
    return m


def ref_step(m, u, i, y):
    """models/MF.py:64-68 verbatim call order (ratings as float32, data/generators.py:129)."""
    m.optimizer.zero_grad()
    loss = m.process_one_batch(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(y))
    loss.backward()
    gP = m.user_embedding.weight.grad.detach().numpy().copy()
    gQ = m.item_embedding.weight.grad.detach().numpy().copy()
    m.optimizer.step()
    return float(loss), gP, gQ


def run_case(name, U, I, d, batches, optimizer, lr, loss_func, seed):
    rng = np.random.default_rng(seed)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    m = make_ref(U, I, d, P0, Q0, optimizer, lr, loss_func)
    orc = G.oracle.MFOracle(P0, Q0, optimizer=optimizer, lr=lr)
    losses, g1 = [], None
    for t, (u, i, y) in enumerate(batches):
        loss, gP, gQ = ref_step(m, u, i, y)
        if t == 0:
            g1 = (gP, gQ)
            ogP, ogQ, _ = orc.pointwise_grad(u, i, y, loss_func)
            assert G.rel_err(ogP, gP) < 2e-6 and G.rel_err(ogQ, gQ) < 2e-6, "oracle grad != reference"
        lo = orc.pointwise_step(u, i, y, loss_func)
        assert abs(lo - loss) < 1e-5 * max(1, abs(loss)), (lo, loss)
        losses.append(loss)
    PT = m.user_embedding.weight.detach().numpy().copy()
    QT = m.item_embedding.weight.detach().numpy().copy()
    eP, eQ = G.rel_err(orc.P, PT), G.rel_err(orc.Q, QT)
    dP = np.abs(PT - P0).max() / np.abs(PT).max()
    print(f"{name}: T={len(batches)} loss0={losses[0]:.6f} lossT={losses[-1]:.6f} oracle rel err P {eP:.2e} Q {eQ:.2e}; "
          f"update / table {dP:.3f}")
    assert max(eP, eQ) < 1e-5
    lens = np.array([len(b[0]) for b in batches], dtype=np.int32)
    cat = lambda k, dt: np.concatenate([b[k] for b in batches]).astype(dt)
    np.savez_compressed(os.path.join(G.OUT, name + ".npz"), P0=P0, Q0=Q0, PT=PT, QT=QT, gP1=g1[0], gQ1=g1[1],
                        u=cat(0, np.int32), i=cat(1, np.int32), y=cat(2, np.float32), batch_len=lens,
                        loss=np.array(losses, dtype=np.float64), lr=np.float32(lr), optimizer=np.array(optimizer),
                        loss_func=np.array(loss_func))


def random_batches(rng, U, I, n, T, ratings):
    """(user, item, rating) with repeated users and items inside a batch"""
    return [(rng.integers(0, U, n).astype(np.int64), rng.integers(0, I, n).astype(np.int64),
             rng.choice(ratings, n).astype(np.float32)) for _ in range(T)]


def generator_batches(U, I, density, batch_size, T, seed):
    """the reference's own PointwiseGenerator on a random implicit matrix (every user has >= 1 positive)"""
    rng = np.random.default_rng(seed)
    dense = (rng.random((U, I)) < density)
    dense[np.arange(U), rng.integers(0, I, U)] = True
    mat = sp.csr_matrix(dense.astype(np.float32))
    G.set_random_seed(2020)
    gen = PointwiseGenerator(mat, return_rating=True, num_negatives=1, batch_size=batch_size, shuffle=True,
                             device=torch.device("cpu"))
    out = []
    for b, (u, i, y) in enumerate(gen):
        out.append((u.numpy().astype(np.int64), i.numpy().astype(np.int64), y.numpy().astype(np.float32)))
        if len(out) == T:
            break
    # what a batch is made of (documents the generator's quirk for the tests): batch_size interactions, then ONE
    # negative for every user of the matrix, whatever users the batch holds
    assert all(len(b[0]) == batch_size + U for b in out[:-1])
    assert all((b[2][batch_size:] == 0).all() and (b[2][:batch_size] == 1).all() for b in out[:-1])
    return out, mat


def main():
    G.oracle.build()
    rng = np.random.default_rng(808)
    run_case("g8_pointwise_ce_sgd_300x200_d32", 300, 200, 32, random_batches(rng, 300, 200, 256, 12, [0.0, 1.0]),
             "sgd", 20.0, "ce", 31)
    run_case("g8_pointwise_mse_sgd_200x150_d64", 200, 150, 64, random_batches(rng, 200, 150, 300, 12, [0.0, 1.0, 3.0, 5.0]),
             "sgd", 2.0, "mse", 32)
    run_case("g8_pointwise_ce_adam_250x120_d128", 250, 120, 128, random_batches(rng, 250, 120, 200, 12, [0.0, 1.0]),
             "adam", 1e-3, "ce", 33)
    gb, mat = generator_batches(120, 90, 0.08, 64, 10, 34)
    run_case("g8_pointwise_ce_adam_generator_120x90_d32", 120, 90, 32, gb, "adam", 1e-3, "ce", 35)
    np.savez_compressed(os.path.join(G.OUT, "g8_pointwise_generator_matrix.npz"), indptr=mat.indptr.astype(np.int64),
                        indices=mat.indices.astype(np.int32), shape=np.array(mat.shape))


if __name__ == "__main__":
    main()
