"""Runner with the flow of the reference's main.py (main.py:29-68): dataset -> Evaluator ->
model = getattr(package, name)(dataset, hparams, device) -> model.fit(...) -> print(scores).

    python -m recsys_pytorch_amd.main --data tests/golden/ml100k_csr.npz --model MF --hidden-dim 32

The reference wires its settings through OmegaConf dataclasses + conf/<Model>.yaml and has no
command line (config.py:49-60); the same fields are plain arguments here.
"""
import argparse
import types

import numpy as np
import torch

import recsys_pytorch_amd as pkg


class ConsoleLogger:
    """anything with log_metrics(dict, epoch=int) (models/MF.py:82-84)"""

    def log_metrics(self, metrics, epoch=None):
        print("epoch %3d  " % epoch + "  ".join("%s=%.4f" % (k, float(v)) for k, v in metrics.items()), flush=True)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="CSR fixture (.npz) or a `user item rating timestamp` text file")
    ap.add_argument("--separator", default="\t")
    ap.add_argument("--model", default="MF", choices=["MF", "LightGCN"])
    ap.add_argument("--hidden-dim", type=int, default=50)           # conf/MF.yaml
    ap.add_argument("--num-layers", type=int, default=2)            # conf/LightGCN.yaml
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adam"])
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--batch-size", type=int, default=256)          # config.py:39
    ap.add_argument("--num-epochs", type=int, default=10)           # config.py:38
    ap.add_argument("--ks", type=int, nargs="+", default=[5])       # config.py:28
    ap.add_argument("--seed", type=int, default=2020)               # config.py:46
    ap.add_argument("--test-step", type=int, default=1)
    args = ap.parse_args(argv)

    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if args.data.endswith(".npz"):
        dataset = pkg.InteractionData.from_npz(args.data)
    else:
        from recsys_pytorch_amd.data import load_uirt
        dataset = load_uirt(args.data, args.separator, min_item_per_user=10, min_user_per_item=1)   # config.py:13-14
    dataset.dataname = "data"
    device = torch.device("cuda")       # the HIP path has no CPU fallback
    evaluator = pkg.Evaluator(dataset.valid_input, dataset.valid_target, protocol=dataset.protocol, ks=args.ks)
    if args.model == "MF":
        hparams = {"hidden_dim": args.hidden_dim, "pointwise": False, "loss_func": "ce", "optimizer": args.optimizer,
                   "seed": args.seed}
    else:
        hparams = {"emb_dim": args.hidden_dim, "num_layers": args.num_layers, "node_dropout": 0.0, "split": False,
                   "num_folds": 100, "reg": 1e-4, "graph_dir": "graph", "seed": args.seed}
    if args.lr is not None:
        hparams["lr"] = args.lr
    model = getattr(pkg, args.model)(dataset, hparams, device)      # main.py:46-47,65
    exp_config = types.SimpleNamespace(batch_size=args.batch_size, num_epochs=args.num_epochs, verbose=0,
                                       test_from=1, test_step=args.test_step)
    ret = model.fit(dataset, exp_config, evaluator=evaluator, loggers=[ConsoleLogger()])
    print(ret["scores"])                                            # main.py:68
    return ret


if __name__ == "__main__":
    main()
