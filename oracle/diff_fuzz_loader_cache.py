"""oracle/diff_fuzz_loader_cache.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container (imports the reference from /root/reference).
The reference writes the cache, data.load_uirt reads it: the same train / valid / test matrices (values too) as the reference holds in memory."""
import os, random, shutil, sys
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(1, "/root/repo"); sys.path.insert(2, "/root/repo/tests")
from data.dataset import UIRTDataset
from recsys_pytorch_amd.data import load_uirt
from loader_cases import case
bad = 0
for seed in list(range(0, 80)) + [133, 192, 268]:
    text, kw = case(seed)
    work = "/tmp/rsx_diff_fuzz/cross"; shutil.rmtree(work, ignore_errors=True); os.makedirs(work)
    path = os.path.join(work, "d.data"); open(path, "w").write(text)
    random.seed(7); np.random.seed(7)
    try:
        ref = UIRTDataset(data_path=path, generalization="weak", **kw)
    except Exception:
        continue
    mine = load_uirt(path, seed=None, cache_dir="cache", **kw)          # served from the reference's cache
    for name, a, b in (("train", ref.train_data, mine.train_data), ("valid", ref.valid_target, mine.valid_target), ("test", ref.test_target, mine.test_target)):
        if a.shape != b.shape or (a != b).nnz:
            bad += 1; print("seed", seed, name, a.shape, b.shape)
    if (ref.num_users, ref.num_items) != (mine.train_data.shape):
        bad += 1; print("seed", seed, "counts", ref.num_users, ref.num_items, mine.train_data.shape)
print("bad", bad)
