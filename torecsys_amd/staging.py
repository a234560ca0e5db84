"""Host -> device staging of index batches.

The reference's collate step turns a list of per-sample field values into ``torch.Tensor(values).long().to(device)``
(data/dataloader/collate_fn.py:65-96: through float32, pageable memory, a blocking copy per field) and
``Inputs.forward`` then ``unsqueeze``s and ``cat``s the fields on the device (inputs/inputs.py:75-80).
``IndexStager`` writes the fields of a batch straight into one pinned ``(B, N)`` buffer -- exact integers, int32 on
the wire when the ids fit (the kernels consume int32 indices natively, so PCIe and HBM traffic for the indices
halves) -- and issues one asynchronous copy on a copy stream; a ring of buffers lets the copy of batch k+1 overlap
the step of batch k.  The device-side counterpart for columns that are already on the GPU is
``functional.pack_columns`` (used by ``Inputs.forward``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

__all__ = ["IndexStager", "pack_host"]

Columns = Union[np.ndarray, torch.Tensor, Sequence, Dict[str, object]]


def pack_host(columns: Columns, out: np.ndarray, names: Optional[Sequence[str]] = None) -> np.ndarray:
    """Write ``columns`` into the 2-D host array ``out`` (B, N), checking shape and integer range.

    ``columns``: a (B, N) array / CPU tensor / nested list, or a sequence of N per-field columns (each of length B or
    shape (B, k)), or a dict of such columns (taken in ``names`` order, default: the dict's order)."""
    B, N = out.shape
    if isinstance(columns, dict):
        keys = list(names) if names is not None else list(columns.keys())
        columns = [columns[k] for k in keys]
    if isinstance(columns, torch.Tensor):
        columns = columns.numpy()
    if isinstance(columns, np.ndarray) and columns.ndim == 2:
        cols = [columns]
    else:
        cols = [c.numpy() if isinstance(c, torch.Tensor) else np.asarray(c) for c in columns]
    info = np.iinfo(out.dtype)
    c0 = 0
    for j, c in enumerate(cols):
        if c.dtype.kind not in "iu":
            if c.dtype.kind == "f" and np.all(c == np.floor(c)):
                c = c.astype(np.int64)
            else:
                raise TypeError(f"column {j}: indices must be integers, got {c.dtype}")
        if c.ndim == 1:
            c = c[:, None]
        if c.ndim != 2 or c.shape[0] != B:
            raise ValueError(f"column {j}: expected {B} samples, got shape {tuple(c.shape)}")
        w = c.shape[1]
        if c0 + w > N:
            raise ValueError(f"columns are wider than the staging buffer ({N} fields)")
        if c.size and (c.min() < info.min or c.max() > info.max):
            raise OverflowError(f"column {j}: values do not fit {out.dtype}; stage with dtype=torch.int64")
        out[:, c0:c0 + w] = c
        c0 += w
    if c0 != N:
        raise ValueError(f"columns fill {c0} of {N} fields")
    return out


class StagedBatch:
    """A device index batch whose copy may still be in flight on the stager's copy stream."""
    __slots__ = ("_tensor", "_event")

    def __init__(self, tensor: torch.Tensor, event):
        self._tensor, self._event = tensor, event

    def wait(self) -> torch.Tensor:
        """Make the CURRENT stream wait for the copy (no host block) and return the (B, N) device tensor."""
        if self._event is not None:
            cur = torch.cuda.current_stream(self._tensor.device.index)
            cur.wait_event(self._event)
            self._tensor.record_stream(cur)
            self._event = None
        return self._tensor


class IndexStager:
    def __init__(self, batch_size: int, num_fields: int, device, dtype: torch.dtype = torch.int32, depth: int = 2,
                 names: Optional[Sequence[str]] = None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("IndexStager stages onto a HIP device (there is no CPU path)")
        if dtype not in (torch.int32, torch.int64):
            raise TypeError(f"IndexStager: dtype {dtype} (int32 or int64)")
        self.device, self.dtype, self.names = device, dtype, list(names) if names is not None else None
        self._host = [torch.empty(batch_size, num_fields, dtype=dtype, pin_memory=True) for _ in range(depth)]
        self._np = [h.numpy() for h in self._host]
        self._done: List[Optional[torch.cuda.Event]] = [None] * depth
        self._k = 0
        self._stream = torch.cuda.Stream(device=device)

    def stage(self, columns: Columns) -> StagedBatch:
        k = self._k
        self._k = (k + 1) % len(self._host)
        if self._done[k] is not None:
            self._done[k].synchronize()          # the copy that last used this pinned buffer must have finished
        pack_host(columns, self._np[k], self.names)
        prev = torch.cuda.current_stream(self.device.index if self.device.index is not None else torch.cuda.current_device())
        torch.cuda.set_stream(self._stream)
        try:
            dev = self._host[k].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        finally:
            torch.cuda.set_stream(prev)
        self._done[k] = ev
        return StagedBatch(dev, ev)
