"""oracle/gen_golden_loader_random.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

Runs the reference's own UIRTDataset (data/dataset.py:92-199, data/preprocess.py:12-90) on the random files of tests/loader_cases.py
and records, per case, the cache directory it wrote and the sha256 of its five files -- or the exception it raised -- into
tests/golden/g10_loader_random.json (data only).  recsys_pytorch_amd/data.py:load_uirt must write the same bytes / fail alike.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_loader_random.py
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
sys.path.insert(1, os.path.join(REPO, "tests"))

from data.dataset import UIRTDataset  # noqa: E402  (reference)
from loader_cases import SEEDS, case  # noqa: E402


def main():
    out = {}
    for seed in SEEDS:
        text, kw = case(seed)
        work = "/tmp/rsx_golden_loader/case"
        shutil.rmtree("/tmp/rsx_golden_loader", ignore_errors=True)
        os.makedirs(work)
        path = os.path.join(work, "d.data")
        open(path, "w").write(text)
        random.seed(7); np.random.seed(7)          # utils/general.py:31-38 via main.py:30
        rec = {"params": {k: v for k, v in kw.items()}, "input_sha256": hashlib.sha256(text.encode()).hexdigest()}
        try:
            ds = UIRTDataset(data_path=path, generalization="weak", **kw)
            rec["shape"] = [int(ds.num_users), int(ds.num_items)]
            rec["matrices"] = {}
            for part, m in (("train", ds.train_data), ("valid", ds.valid_target), ("test", ds.test_target)):
                m = m.tocsr().copy(); m.sum_duplicates(); m.sort_indices()
                rec["matrices"][part] = hashlib.sha256(m.indptr.astype(np.int64).tobytes() + m.indices.astype(np.int64).tobytes()
                                                       + m.data.astype(np.float64).tobytes()).hexdigest()
            (sub,) = os.listdir(os.path.join(work, "cache"))
            rec["cache_subdir"] = sub
            rec["files"] = {name: hashlib.sha256(open(os.path.join(work, "cache", sub, name), "rb").read()).hexdigest()
                            for name in sorted(os.listdir(os.path.join(work, "cache", sub)))}
        except Exception as e:      # noqa: BLE001 -- what the reference raises IS the expected outcome
            rec["raises"] = type(e).__name__
        out[str(seed)] = rec
    json.dump(out, open(os.path.join(REPO, "tests", "golden", "g10_loader_random.json"), "w"), indent=1, sort_keys=True)
    print({k: (v.get("raises") or v["cache_subdir"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
