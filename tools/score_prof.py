"""tools/score_prof.py [tiles] : one fused scoring call (the thing rocprofv3 wraps in tools/pmc_score.sh)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_pytorch_amd import rsx
from recsys_pytorch_amd.data import synthetic_csr
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
U, I, d, K = 200_000, 100_000, 128, 50
dev = torch.device("cuda")
torch.manual_seed(0)
P = torch.randn(U, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
ip, ix = synthetic_csr(U, I, 20, dev)
users = torch.arange(1024 * tiles, device=dev, dtype=torch.int32)
rsx.set_option("score_lanes", int(os.environ.get("LANES", 1)))
if os.environ.get("ABL"):      # development write switches: needs RSX_LIB=.../librsx_dev.so
    rsx.lib().rsx_debug_set_score_ablation(int(os.environ["ABL"]))
for _ in range(2):
    top = rsx.score_topk(P, Q, users, K, mask=(ip, ix))
torch.cuda.synchronize()
