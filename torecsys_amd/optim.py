"""Fused sparse optimizers for embedding tables.

The reference trains every parameter with a dense optimizer over a dense ``V x E`` gradient
(trainer/torecsys_pipeline.py:562-578).  For SGD and Adagrad (no momentum, no weight decay) the dense step only
changes rows that were looked up, so it can be applied by the kernel that reduces the bucketed gradient
(``FusedSparseAdam`` applies the lazy ``torch.optim.SparseAdam`` rule instead -- dense Adam also decays the moments
of rows nobody looked up, which is exactly the full-table sweep a 1 B-row table cannot afford):
``module.set_fused_optimizer(FusedSparseSGD(lr))`` makes the backward pass update the table rows in place --
bit-for-bit the same rows a dense ``torch.optim.SGD`` / ``Adagrad`` step would produce up to fp32 summation order --
without ever forming the dense gradient or sweeping the table.  The other parameters keep a normal optimizer.

State (Adagrad accumulators, Adam moments and step counts) is kept per table, keyed by the ``nn.Parameter`` object, so
it follows ``module.to(...)`` (the state tensors move to the table's device on the next step) and can be saved /
restored with ``state_dict(named_parameters)`` / ``load_state_dict(sd, named_parameters)``.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch


class _FusedSparse:
    kind = 0
    _buffers: Tuple[str, ...] = ()          # names of the per-table fp32 state tensors

    def __init__(self, lr: float, eps: float = 0.0):
        if lr < 0:
            raise ValueError(f"invalid learning rate {lr}")
        self.lr, self.eps = float(lr), float(eps)
        self._state: Dict[object, dict] = {}

    # ---- per-table state -------------------------------------------------------------------------------
    def _init_value(self, name: str) -> float:
        return 0.0

    def _entry(self, table: torch.Tensor, key=None) -> dict:
        """state of one table.  ``key`` is the nn.Parameter the rows belong to (falls back to the storage address for
        bare tensors); tensors follow the table's device and are rebuilt if its shape changed."""
        k = id(key) if key is not None else (table.data_ptr(), tuple(table.shape))
        st = self._state.get(k)
        if st is None:
            st = self._state[k] = {"step": 0, "ref": key}
        for name in self._buffers:
            t = st.get(name)
            if t is None or tuple(t.shape) != tuple(table.shape):
                st[name] = torch.full(table.shape, self._init_value(name), dtype=torch.float32, device=table.device)
            elif t.device != table.device:
                st[name] = t.to(table.device)
        return st

    def state_for(self, table: torch.Tensor, key=None):
        return None

    # ---- checkpointing ---------------------------------------------------------------------------------
    def state_dict(self, named_parameters: Optional[Iterable] = None) -> dict:
        """{'hyper': {...}, 'tables': {name: {'step': int, <buffer>: tensor}}}; ``named_parameters`` (e.g.
        ``model.named_parameters()``) gives the tables their names -- tables it does not cover are 'table<i>'."""
        names = {id(p): n for n, p in (named_parameters or [])}
        tables = {}
        for i, (k, st) in enumerate(self._state.items()):
            name = names.get(k if isinstance(k, int) else None, f"table{i}")
            tables[name] = {"step": st["step"], **{b: st[b].detach().clone() for b in self._buffers if b in st}}
        return {"hyper": {k: v for k, v in self.__dict__.items() if isinstance(v, (int, float))}, "tables": tables}

    def load_state_dict(self, sd: dict, named_parameters: Iterable) -> None:
        params = dict(named_parameters)
        for k, v in sd.get("hyper", {}).items():
            setattr(self, k, v)
        for name, entry in sd.get("tables", {}).items():
            if name not in params:
                raise KeyError(f"load_state_dict: no parameter named {name!r}")
            p = params[name]
            st = self._state[id(p)] = {"step": int(entry.get("step", 0)), "ref": p}
            for b in self._buffers:
                if b in entry:
                    if tuple(entry[b].shape) != tuple(p.shape):
                        raise ValueError(f"load_state_dict: state {b!r} of {name!r} has shape {tuple(entry[b].shape)}, "
                                         f"the parameter has {tuple(p.shape)}")
                    st[b] = entry[b].to(device=p.device, dtype=torch.float32).clone()


class FusedSparseSGD(_FusedSparse):
    """w[r] -= lr * g[r] for every looked-up row r (== torch.optim.SGD(lr) without momentum / weight decay)."""
    kind = 1


class FusedSparseAdagrad(_FusedSparse):
    """state[r] += g[r]^2 ; w[r] -= lr * g[r] / (sqrt(state[r]) + eps)   (== torch.optim.Adagrad(lr, eps=eps))."""
    kind = 2
    _buffers = ("sum",)

    def __init__(self, lr: float = 1e-2, eps: float = 1e-10, initial_accumulator_value: float = 0.0):
        super().__init__(lr, eps)
        self.initial = float(initial_accumulator_value)

    def _init_value(self, name: str) -> float:
        return self.initial

    def state_for(self, table: torch.Tensor, key=None):
        return self._entry(table, key)["sum"]


class FusedSparseAdam(_FusedSparse):
    """Lazy Adam on the looked-up rows, == ``torch.optim.SparseAdam(lr, betas, eps)`` on the coalesced sparse gradient:
    m[r] += (g[r]-m[r])(1-b1);  v[r] += (g[r]^2-v[r])(1-b2);  w[r] -= lr*sqrt(1-b2^t)/(1-b1^t) * m[r]/(sqrt(v[r])+eps).
    ``t`` counts the steps applied to each table (one per backward pass through it)."""
    kind = 3
    _buffers = ("exp_avg", "exp_avg_sq")

    def __init__(self, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(lr, eps)
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid betas {betas}")
        self.beta1, self.beta2 = float(betas[0]), float(betas[1])

    def state_for(self, table: torch.Tensor, key=None):
        st = self._entry(table, key)
        return st["exp_avg"], st["exp_avg_sq"]

    def next_step_size(self, table: torch.Tensor, key=None) -> float:
        st = self._entry(table, key)
        st["step"] += 1
        t = st["step"]
        return self.lr * (1.0 - self.beta2 ** t) ** 0.5 / (1.0 - self.beta1 ** t)
