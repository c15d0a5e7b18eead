"""oracle/gen_golden_odd_dims.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

G1c (SGD at a resolvable step size, repeated users and items) at two hidden_dim values far from the kernels' row widths: 1 (stored as 32
columns, 31 of them the zero pad) and 77 (stored as 128), produced by the reference's own loss / backward / optimizer like the other
fixtures (oracle/gen_golden.py: run_case, which asserts oracle == reference while generating).  Round 5.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_odd_dims.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)


def main():
    G.oracle.build()
    rng = np.random.default_rng(77)
    G.run_case("g1c_sgd_biglr_150x90_d1_b64", 150, 90, 1, G.random_batches(rng, 150, 90, 64, 12), "sgd", 6.0, 41)
    G.run_case("g1c_sgd_biglr_220x130_d77_b128", 220, 130, 77, G.random_batches(rng, 220, 130, 128, 12), "sgd", 10.0, 42)
    for name in ("g1c_sgd_biglr_150x90_d1_b64", "g1c_sgd_biglr_220x130_d77_b128"):
        z = np.load(os.path.join(G.OUT, name + ".npz"))
        for t in ("P", "Q"):
            print(f"  {name}: max|{t}T-{t}0| / max|{t}T| = {np.abs(z[t + 'T'] - z[t + '0']).max() / np.abs(z[t + 'T']).max():.3f}")


if __name__ == "__main__":
    main()
