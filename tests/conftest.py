import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def split_batches(g):
    """yield (u, i, j) int64 batches from a g1*/g1b* fixture."""
    off = 0
    for n in g["batch_len"]:
        n = int(n)
        yield (g["u"][off:off + n].astype(np.int64), g["i"][off:off + n].astype(np.int64),
               g["j"][off:off + n].astype(np.int64))
        off += n


def rel_err(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))
                 / (np.max(np.abs(b)) + 1e-30))


G1_SGD = ["g1_sgd_200x100_d32_b64", "g1_sgd_ml100k_d32_b256",
          "g1_sgd_500x300_d64_b257", "g1_sgd_400x250_d128_b512"]
G1_ADAM = ["g1b_adam_200x100_d32_b64", "g1b_adam_ml100k_d32_b256"]
G23 = [n.replace("g1_sgd", "g23") for n in G1_SGD]


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build(with_ref=os.path.exists("/root/reference"))
    return oracle
