"""oracle/gen_golden_shipped_config.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

The reference exactly as it ships (BASELINE configs[0]): conf/MF.yaml's hidden_dim 50, the optimizer of models/MF.py:30 (Adam, lr 1e-3), the
ml-100k split of its own loader and the first 12 batches of its own PairwiseGenerator (taken from the fixture g1b_adam_ml100k_d32_b256,
which oracle/gen_golden.py recorded from that generator) -> tests/golden/g1b_adam_ml100k_d50_b256.npz, through oracle/gen_golden.py's
run_case (which asserts oracle == reference while generating).  Round 5.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_shipped_config.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)


def main():
    G.oracle.build()
    z = np.load(os.path.join(G.OUT, "g1b_adam_ml100k_d32_b256.npz"))
    U, I = z["P0"].shape[0], z["Q0"].shape[0]
    cuts = np.concatenate([[0], np.cumsum(z["batch_len"])])
    batches = [tuple(z[k][cuts[t]:cuts[t + 1]].astype(np.int64) for k in ("u", "i", "j")) for t in range(len(z["batch_len"]))]
    G.run_case("g1b_adam_ml100k_d50_b256", U, I, 50, batches, "adam", 1e-3, 61)


if __name__ == "__main__":
    main()
