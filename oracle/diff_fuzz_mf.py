"""oracle/diff_fuzz_mf.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (imports the reference from /root/reference).
Differential run of the reference's own code against this repository's restatement on random inputs (round 5; results:
profiles/r05_fuzz_campaign.txt).  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/diff_fuzz_mf.py <first seed> <last seed>"""
import os, sys, types
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(1, "/root/repo")
np.int = int; np.float = float
import torch
torch.set_num_threads(1)
from models.MF import MF
import oracle
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    U, I = int(rng.integers(1, 400)), int(rng.integers(2, 300))
    d = int(rng.integers(1, 257))
    B = int(rng.integers(1, 700))
    pointwise = seed % 4 == 3
    lf = "mse" if seed % 8 == 7 else "ce"
    opt = "adam" if seed % 2 else "sgd"
    lr = 1e-3 if opt == "adam" else 0.05 * B * 0.2
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    ds = types.SimpleNamespace(num_users=U, num_items=I)
    m = MF(ds, {"hidden_dim": d, "pointwise": pointwise, "loss_func": lf}, torch.device("cpu"))
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(P0)); m.item_embedding.weight.copy_(torch.from_numpy(Q0))
    if opt == "sgd":
        m.optimizer = torch.optim.SGD(m.parameters(), lr=lr)
    orc = oracle.MFOracle(P0, Q0, opt, lr)
    ctx = f"seed {seed}: U={U} I={I} d={d} B={B} {opt} pointwise={pointwise} {lf}"
    ok = True
    for t in range(3):
        u, i = rng.integers(0, U, B), rng.integers(0, I, B)
        third = (rng.integers(0, 2, B) if lf == "ce" else rng.integers(1, 6, B)).astype(np.float32) if pointwise else rng.integers(0, I, B)
        m.optimizer.zero_grad()
        loss = m.process_one_batch(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(third))
        loss.backward()
        gP = m.user_embedding.weight.grad.detach().numpy().copy(); gQ = m.item_embedding.weight.grad.detach().numpy().copy()
        # the oracle's gradient at the REFERENCE's current tables
        o2 = oracle.MFOracle(m.user_embedding.weight.detach().numpy(), m.item_embedding.weight.detach().numpy(), "sgd", 1.0)
        ogP, ogQ, ol = (o2.pointwise_grad(u, i, third, lf) if pointwise else o2.grad(u, i, third))
        eg = max(np.abs(ogP - gP).max() / (np.abs(gP).max() + 1e-30), np.abs(ogQ - gQ).max() / (np.abs(gQ).max() + 1e-30))
        m.optimizer.step()
        lo = orc.pointwise_step(u, i, third, lf) if pointwise else orc.step(u, i, third)
        if eg > 5e-6 or abs(ol - float(loss)) > 2e-6 * max(1, abs(float(loss))):
            ok = False; print(ctx, "step", t, "grad err", eg, "loss", ol, float(loss))
    PT, QT = m.user_embedding.weight.detach().numpy(), m.item_embedding.weight.detach().numpy()
    upd = max(np.abs(PT - P0).max(), np.abs(QT - Q0).max(), 1e-30)
    e = max(np.abs(orc.P - PT).max(), np.abs(orc.Q - QT).max()) / upd
    tol = 2e-3 if opt == "adam" else 2e-5
    if e > tol:
        ok = False; print(ctx, "tables: error / update", e)
    bad += not ok
print("bad", bad)
