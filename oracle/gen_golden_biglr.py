"""oracle/gen_golden_biglr.py -- TEST INFRASTRUCTURE.  Runs ONLY in the build container.

G1c: the G1 step-parity fixtures again with a LARGE learning rate, so that 20 steps move the
tables by O(1) of their magnitude.  With the G1 fixtures (lr 0.05, gradients carrying 1/B) the
whole 20-step update is only ~1e-3 of max|table|, so an assertion "tables within 1e-5 relative"
resolves the UPDATE to a percent only; here the same assertion resolves it to ~1e-5.  Produced by
the reference's own loss/backward with the SGD-swapped optimizer, exactly like G1
(oracle/gen_golden.py:run_case, which also asserts oracle == reference while generating).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_biglr.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (imports the reference read-only)


def main():
    G.oracle.build()
    rng = np.random.default_rng(101)
    for name, U, I, d, B, lr, seed in (("g1c_sgd_biglr_200x100_d32_b64", 200, 100, 32, 64, 20.0, 21),
                                       ("g1c_sgd_biglr_400x250_d128_b512", 400, 250, 128, 512, 40.0, 22)):
        G.run_case(name, U, I, d, G.random_batches(rng, U, I, B, 20), "sgd", lr, seed)
        z = np.load(os.path.join(G.OUT, name + ".npz"))
        for t in ("P", "Q"):
            d_ = np.abs(z[t + "T"] - z[t + "0"]).max() / np.abs(z[t + "T"]).max()
            print(f"  {name}: max|{t}T-{t}0| / max|{t}T| = {d_:.3f}")


if __name__ == "__main__":
    main()
