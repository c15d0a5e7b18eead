"""tools/step_prof.py [B] [neg_block] [steps] : the native loop on the bench workload for a few steps -- the thing
rocprofv3 wraps in tools/pmc_groups.py (no timing of its own).  Shape through the environment: USERS, ITEMS, DIM, DEG, POP.
With STEP_PROF_META=<file> it also writes which bench leg this is: the traffic key bench.py looks up in
profiles/traffic.json and the step kernel's full name (tools/install_profiles.py reads it)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
from recsys_pytorch_amd.sharded import BPREngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nbw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
U, I, d = int(os.environ.get("USERS", 1_000_000)), int(os.environ.get("ITEMS", 100_000)), int(os.environ.get("DIM", 128))
pop = os.environ.get("POP", "zipf")
dev = torch.device("cuda")
torch.manual_seed(2020)
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, int(os.environ.get("DEG", 20)), dev, popularity=pop)
eng = BPREngine(P, Q, 0.05)
nb = eng.set_neg_block(B, nbw) if nbw else 0
if nb and os.environ.get("NEG_EXACT"):        # the block size itself (experiments)
    nb = eng.neg_block = int(os.environ["NEG_EXACT"]); eng._csr = None
eng.set_hot_items(torch.bincount(ix.long(), minlength=I), 256)
ck = int(os.environ.get("CHUNKS", 0))
if ck > 1 and (nb or nbw):      # (B < 2 I: the ranges without blocks)
    eng.set_chunks(ck)
    nb = eng.neg_block           # (ranges use blocks of at least 3)
    if os.environ.get("NEG_EXACT"):
        nb = eng.neg_block = int(os.environ["NEG_EXACT"]); eng._csr = None; eng._relabel = None
if os.environ.get("STEP_PROF_META"):
    kernel = (f"bpr_step_blocked_kernel<{d}, 3, unsigned int, {'true' if nb else 'false'}>" if (nb or eng._sorts(B))
              else f"bpr_step_kernel<{d}, 0, 3, unsigned int>")
    import bench                  # (repo root is on sys.path) the hash of the kernel sources THIS run measures: bench.py's stale flag
    json.dump({"key": f"U{U}_I{I}_d{d}_B{B}_{pop}_nb{nb}" + (f"_c{ck}" if (ck > 1 and (nb or nbw)) else ""), "kernel": kernel, "argv": sys.argv[1:],
               "sources_sha": bench.sources_sha("step"),
               "env": {k: os.environ[k] for k in ("USERS", "ITEMS", "DIM", "DEG", "POP", "CHUNKS") if k in os.environ}},
              open(os.environ["STEP_PROF_META"], "w"))
loss = torch.zeros(rsx.RSX_LOSS_SLOTS, device=dev)
tr = eng.native_trainer(ip, ix, B, loss_acc=loss)
tr.run(steps)
torch.cuda.synchronize()
