"""oracle/torch_port.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PyTorch-CPU port of the reference's MF hot path, op for op, so that its COST
PROFILE (dense [U x d] + [I x d] gradients, optimizer sweep over every row) is
the reference's.  This is what ``bench.py`` times as ``cpu_baseline`` (kind
"port") on the GPU box's host cores, where /root/reference does not exist.
Validated against the imported reference in tests/golden (gen_golden.py runs
both and asserts agreement before writing fixtures).

Follows:
  models/MF.py:23-24,30   two nn.Embedding tables, Adam(lr=1e-3) as shipped
  models/MF.py:32-42      gather + sum(mul(.,.), 1)
  models/MF.py:99-107     -sigmoid(pos - neg).log().mean()
  models/MF.py:64-68      zero_grad / backward / step
  models/MF.py:109-112    user_latent @ item_table.T
"""
import numpy as np
import torch


class TorchMFPort(torch.nn.Module):
    def __init__(self, P0, Q0, optimizer="adam", lr=1e-3):
        super().__init__()
        P0 = torch.as_tensor(np.asarray(P0), dtype=torch.float32)
        Q0 = torch.as_tensor(np.asarray(Q0), dtype=torch.float32)
        self.user_embedding = torch.nn.Embedding(P0.shape[0], P0.shape[1])
        self.item_embedding = torch.nn.Embedding(Q0.shape[0], Q0.shape[1])
        with torch.no_grad():
            self.user_embedding.weight.copy_(P0)
            self.item_embedding.weight.copy_(Q0)
        if optimizer == "adam":
            self.optimizer = torch.optim.Adam(self.parameters(), lr=lr)
        elif optimizer == "sgd":
            self.optimizer = torch.optim.SGD(self.parameters(), lr=lr)
        else:
            raise ValueError(optimizer)

    def rating(self, users, items):
        return (self.user_embedding(users) * self.item_embedding(items)).sum(1)

    def bpr_loss(self, users, pos, neg):
        x = self.rating(users, pos) - self.rating(users, neg)
        return -torch.sigmoid(x).log().mean()

    def step(self, users, pos, neg):
        users, pos, neg = (torch.as_tensor(np.asarray(a), dtype=torch.long) for a in (users, pos, neg))
        self.optimizer.zero_grad()
        loss = self.bpr_loss(users, pos, neg)
        loss.backward()
        self.optimizer.step()
        return float(loss)

    @torch.no_grad()
    def score(self, users):
        users = torch.as_tensor(np.asarray(users), dtype=torch.long)
        return (self.user_embedding(users) @ self.item_embedding.weight.T).numpy()

    @property
    def P(self):
        return self.user_embedding.weight.detach().numpy()

    @property
    def Q(self):
        return self.item_embedding.weight.detach().numpy()


def topk_numpy(scores, K):
    """numpy twin of the top-k (evaluation/backend/python/func.py:4-17):
    argpartition of -scores, then argsort of the partition."""
    part = np.argpartition(-scores, K, axis=1)[:, :K]
    vals = np.take_along_axis(scores, part, 1)
    order = np.argsort(-vals, axis=1)
    return np.take_along_axis(part, order, 1)
