"""oracle/gen_golden_d256.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

The reference takes any hidden_dim (models/MF.py:19,23-24); the HIP kernels are instantiated for d in {32, 64, 128, 256} (other
values are zero-padded columns).  G1 / G1c / G1b / G23 once more at d = 256, and at the un-padded hidden_dim = 200 (which the
product pads to 256), produced by the reference's own loss / backward / optimizer exactly like the other fixtures
(oracle/gen_golden.py: run_case and g23_case, which assert oracle == reference while generating).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_d256.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)


def main():
    G.oracle.build()
    rng = np.random.default_rng(256)
    m = G.run_case("g1_sgd_300x180_d256_b300", 300, 180, 256, G.random_batches(rng, 300, 180, 300, 12), "sgd", 0.05, 31)
    G.g23_case("g1_sgd_300x180_d256_b300", m)
    G.run_case("g1c_sgd_biglr_300x180_d256_b300", 300, 180, 256, G.random_batches(rng, 300, 180, 300, 12), "sgd", 30.0, 32)
    G.run_case("g1c_sgd_biglr_260x140_d200_b200", 260, 140, 200, G.random_batches(rng, 260, 140, 200, 12), "sgd", 20.0, 33)
    G.run_case("g1b_adam_150x90_d256_b64", 150, 90, 256, G.random_batches(rng, 150, 90, 64, 12), "adam", 1e-3, 34)
    for name in ("g1c_sgd_biglr_300x180_d256_b300", "g1c_sgd_biglr_260x140_d200_b200"):
        z = np.load(os.path.join(G.OUT, name + ".npz"))
        for t in ("P", "Q"):
            d_ = np.abs(z[t + "T"] - z[t + "0"]).max() / np.abs(z[t + "T"]).max()
            print(f"  {name}: max|{t}T-{t}0| / max|{t}T| = {d_:.3f}")


if __name__ == "__main__":
    main()
