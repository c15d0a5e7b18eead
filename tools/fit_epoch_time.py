"""tools/fit_epoch_time.py : what one epoch of MF.fit costs when an epoch IS one step (batch = users, the headline shape) -- the model
class's loop around the native trainer, against bench.py's 50 steps per call.  python tools/fit_epoch_time.py [epochs] [profile]
Every shape twice: the loop as it is, and the loop as it was until round 6 (a seek at every epoch start, which drops the batches sampled
ahead, and the epoch's loss read back every epoch) -- restored here by making the trainer's state compare unequal and handing in a logger."""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import recsys_pytorch_amd as pkg
from recsys_pytorch_amd.data import synthetic_csr
U, I, d = 1_000_000, 100_000, 128
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ip, ix = synthetic_csr(U, I, 20, "cuda", seed=2020)
R = sp.csr_matrix((np.ones(U * 20, np.float32), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
ds = pkg.InteractionData(R)
from recsys_pytorch_amd import rsx
real_state = rsx.BPRTrainer.state


class NeverEqual(tuple):
    __eq__ = lambda self, other: False
    __ne__ = lambda self, other: True
    __hash__ = tuple.__hash__


class NullLogger:
    def log_metrics(self, summary, epoch=None):
        pass


for nb, old in ((8, False), (8, True), (0, False), (0, True)):
    rsx.BPRTrainer.state = (lambda self: NeverEqual(real_state(self))) if old else real_state
    loggers = [NullLogger()] if old else None
    m = pkg.MF(ds, {"hidden_dim": d, "pointwise": False, "loss_func": "ce", "lr": 0.05, "neg_block": nb}, "cuda")
    cfg = lambda n: types.SimpleNamespace(batch_size=U, num_epochs=n, verbose=0, test_from=1, test_step=1)
    m.fit(ds, cfg(20), loggers=loggers); torch.cuda.synchronize()
    t = time.perf_counter(); m.fit(ds, cfg(epochs), loggers=loggers); torch.cuda.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter(); m.fit(ds, cfg(1), loggers=loggers); torch.cuda.synchronize(); d1 = time.perf_counter() - t
    print(f"neg_block {nb} (engine: {m._engine.neg_block}) {'OLD loop' if old else 'loop    '}: {epochs} one-step epochs {dt * 1e3:.1f} ms; a fit of ONE epoch (set-up + step) {d1 * 1e3:.2f} ms "
          f"-> {(dt - d1) / (epochs - 1) * 1e6:.1f} us per further epoch")
    if len(sys.argv) > 2 and nb and not old:
        pr = cProfile.Profile(); pr.enable(); m.fit(ds, cfg(epochs)); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
