"""one rsx_bpr_trainer_run for all steps vs step by step, on the shapes the random-shape test of tests/test_sharded_gloo.py found"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle as oracle_mod
import test_sharded_gloo as T


def main():
    base = dict(world=3, exchange="direct", chunks=3, d=256, I=782, B=1037, U=2364, deg=7)
    variants = [("steps 1", dict(steps=1)), ("steps 1, 2 ranks, allreduce", dict(steps=1, world=2, exchange="allreduce")), ("steps 1 again", dict(steps=1)),
                ("I 5000 steps 1", dict(steps=1, I=5000)), ("I 5000 B 6000 U 9000 steps 3", dict(I=5000, B=6000, U=9000))]
    port = 29500 + os.getpid() % 1500
    for n, (name, ch) in enumerate(variants):
        c = dict(base, steps=3); c.update(ch)
        try:
            T._check_ranges(oracle_mod, c["world"], port + 10 * n, True, c["U"], c["I"], c["d"], c["B"], c["deg"], c["chunks"], c["steps"], exchange=c["exchange"])
            print(f"[{name}] ok", flush=True)
        except AssertionError as e:
            print(f"[{name}] FAILED: {str(e).splitlines()[0][:300]}", flush=True)
        except Exception as e:      # noqa: BLE001
            print(f"[{name}] ERROR: {repr(e)[:300]}", flush=True)


if __name__ == "__main__":
    main()
