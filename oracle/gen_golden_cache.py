"""oracle/gen_golden_cache.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Runs the reference's own UIRTDataset (data/dataset.py:92-199) on a /tmp copy of ml-100k and
records what its on-disk cache looks like: directory name, file names, sizes, sha256 digests and
the first lines of every file -> tests/golden/g7_ml100k_cache.json (data only).  The build's
loader (recsys_pytorch_amd/data.py:load_uirt(cache_dir=...)) must write byte-identical files.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_cache.py
"""
import hashlib
import json
import os
import random
import shutil
import sys

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from data.dataset import UIRTDataset  # noqa: E402  (reference)


def main():
    work = "/tmp/rsx_golden_cache/ml-100k"
    shutil.rmtree("/tmp/rsx_golden_cache", ignore_errors=True)
    os.makedirs(work)
    shutil.copy(os.path.join(REF, "datasets/ml-100k/u.data"), work)
    random.seed(2020); np.random.seed(2020)          # utils/general.py:31-38 via main.py:30
    UIRTDataset(data_path=os.path.join(work, "u.data"), separator="\t", min_item_per_user=10, min_user_per_item=1,
                protocol="holdout", generalization="weak", valid_ratio=0.1, test_ratio=0.2, split_random=True)
    cache_root = os.path.join(work, "cache")
    (sub,) = os.listdir(cache_root)
    out = {"cache_subdir": sub, "files": {}}
    for name in sorted(os.listdir(os.path.join(cache_root, sub))):
        raw = open(os.path.join(cache_root, sub, name), "rb").read()
        out["files"][name] = {"bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest(),
                              "head": raw.decode().splitlines()[:3]}
    json.dump(out, open(os.path.join(REPO, "tests", "golden", "g7_ml100k_cache.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
