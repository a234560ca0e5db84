"""Fused sparse optimizers for embedding tables.

The reference trains every parameter with a dense optimizer over a dense ``V x E`` gradient
(trainer/torecsys_pipeline.py:562-578).  For SGD and Adagrad (no momentum, no weight decay) the dense step only
changes rows that were looked up, so it can be applied by the kernel that reduces the bucketed gradient
(``FusedSparseAdam`` applies the lazy ``torch.optim.SparseAdam`` rule instead -- dense Adam also decays the moments
of rows nobody looked up, which is exactly the full-table sweep a 1 B-row table cannot afford):
``module.set_fused_optimizer(FusedSparseSGD(lr))`` makes the backward pass update the table rows in place --
bit-for-bit the same rows a dense ``torch.optim.SGD`` / ``Adagrad`` step would produce up to fp32 summation order --
without ever forming the dense gradient or sweeping the table.  The other parameters keep a normal optimizer.
"""
from __future__ import annotations

import torch


class _FusedSparse:
    kind = 0

    def __init__(self, lr: float, eps: float = 0.0):
        if lr < 0:
            raise ValueError(f"invalid learning rate {lr}")
        self.lr, self.eps = float(lr), float(eps)

    def state_for(self, table: torch.Tensor):
        return None


class FusedSparseSGD(_FusedSparse):
    """w[r] -= lr * g[r] for every looked-up row r (== torch.optim.SGD(lr) without momentum / weight decay)."""
    kind = 1


class FusedSparseAdagrad(_FusedSparse):
    """state[r] += g[r]^2 ; w[r] -= lr * g[r] / (sqrt(state[r]) + eps)   (== torch.optim.Adagrad(lr, eps=eps))."""
    kind = 2

    def __init__(self, lr: float = 1e-2, eps: float = 1e-10, initial_accumulator_value: float = 0.0):
        super().__init__(lr, eps)
        self.initial = float(initial_accumulator_value)
        self._state = {}

    def state_for(self, table: torch.Tensor):
        key = (table.data_ptr(), tuple(table.shape))
        st = self._state.get(key)
        if st is None:
            st = torch.full(table.shape, self.initial, dtype=torch.float32, device=table.device)
            self._state[key] = st
        return st


class FusedSparseAdam(_FusedSparse):
    """Lazy Adam on the looked-up rows, == ``torch.optim.SparseAdam(lr, betas, eps)`` on the coalesced sparse gradient:
    m[r] += (g[r]-m[r])(1-b1);  v[r] += (g[r]^2-v[r])(1-b2);  w[r] -= lr*sqrt(1-b2^t)/(1-b1^t) * m[r]/(sqrt(v[r])+eps).
    ``t`` counts the steps applied to each table (one per backward pass through it)."""
    kind = 3

    def __init__(self, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(lr, eps)
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid betas {betas}")
        self.beta1, self.beta2 = float(betas[0]), float(betas[1])
        self._state = {}

    def _entry(self, table: torch.Tensor):
        key = (table.data_ptr(), tuple(table.shape))
        st = self._state.get(key)
        if st is None:
            st = self._state[key] = {
                "step": 0,
                "exp_avg": torch.zeros(table.shape, dtype=torch.float32, device=table.device),
                "exp_avg_sq": torch.zeros(table.shape, dtype=torch.float32, device=table.device),
            }
        return st

    def state_for(self, table: torch.Tensor):
        st = self._entry(table)
        return st["exp_avg"], st["exp_avg_sq"]

    def next_step_size(self, table: torch.Tensor) -> float:
        st = self._entry(table)
        st["step"] += 1
        t = st["step"]
        return self.lr * (1.0 - self.beta2 ** t) ** 0.5 / (1.0 - self.beta1 ** t)
