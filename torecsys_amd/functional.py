"""Functional layer: torch.autograd.Functions whose forward/backward are calls into libtrs_hip.so.

Every function takes and returns UN-NAMED tensors on one HIP device (the nn.Modules in
``inputs.py`` / ``layers.py`` strip and re-apply the reference's tensor names).  There is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _abi
from ._abi import call, index_dtype_code, ptr, require_device, size_query, stream_ptr, value_dtype_code

CHECK_INDICES = os.environ.get("TRS_CHECK_INDICES", "0") not in ("", "0")


def _as_index(idx: torch.Tensor) -> torch.Tensor:
    """Indices are consumed as int64 or int32 directly (the reference's ``.long()`` /
    int64-promotion happens inside the kernel); other int dtypes are widened here."""
    idx = idx.rename(None) if idx.has_names() else idx
    if idx.dtype not in (torch.int64, torch.int32):
        if idx.dtype.is_floating_point or idx.dtype == torch.bool:
            raise TypeError(f"indices must be an integer tensor, got {idx.dtype}")
        idx = idx.long()
    return idx.contiguous()


_lazy_flags = {}          # device -> persistent int32[1]: "some lookup was out of range" (never read unless asked)


class _ErrFlag:
    """Out-of-range flag of one lookup.  The kernels range-check every row id when they are handed a flag (an
    out-of-range lookup then reads as a zero row and contributes no gradient instead of touching foreign memory).
    With TRS_CHECK_INDICES=1 the flag is private to the call and read back at once (IndexError, like nn.Embedding);
    otherwise the calls share one persistent flag per device that nobody waits for -- ``index_errors_seen()`` reads it."""

    def __init__(self, dev):
        if CHECK_INDICES:
            self.t, self.lazy = torch.zeros(1, dtype=torch.int32, device=dev), False
        else:
            t = _lazy_flags.get(dev)
            if t is None:
                t = _lazy_flags[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
            self.t, self.lazy = t, True

    def check(self, what):
        if not self.lazy and int(self.t.item()) != 0:
            raise IndexError(f"{what}: index out of range in self")


def index_errors_seen(device=None, reset: bool = True) -> bool:
    """True when a lookup since the last call had a row id outside its table (synchronises).  Only meaningful without
    TRS_CHECK_INDICES=1 (which raises at the offending call instead)."""
    seen = False
    for dev, t in list(_lazy_flags.items()):
        if device is not None and torch.device(device) != dev:
            continue
        if int(t.item()) != 0:
            seen = True
            if reset:
                t.zero_()
    return seen


# --------------------------------------------------------------------------------------------
# Row-bucketed index of one batch (CSR over destination rows), shared by every table looked up
# with the same index tensor (E=64 embeddings and the E=1 first-order weights).
# --------------------------------------------------------------------------------------------
class RowBuckets:
    __slots__ = ("row_start", "perm", "V", "BN", "N", "ready", "stream")

    def __init__(self, row_start, perm, V, BN, N):
        self.row_start, self.perm, self.V, self.BN, self.N = row_start, perm, V, BN, N
        self.ready = None        # event recorded behind the build on the stream it ran on
        self.stream = None       # that stream (side-stream prefetch or the consumer's own stream for an inline build)

    def built_on(self, stream):
        """Record where the build was enqueued: every consumer on ANOTHER stream (the E = 1 table's walk on the "lookup"
        stream sharing the wide table's entry, or the other way round) waits for the event first."""
        self.stream = stream
        self.ready = torch.cuda.Event()
        self.ready.record(stream)

    def wait(self):
        """Make the current stream wait for the build (no-op when built on this stream)."""
        if self.ready is None:
            return
        cur = _abi.current_stream_of(self.row_start.device)
        if self.stream is not None and cur == self.stream:
            return
        cur.wait_event(self.ready)
        self.row_start.record_stream(cur)
        self.perm.record_stream(cur)


_bucket_cache: List[tuple] = []   # [(key, idx_tensor_kept_alive, RowBuckets)]
_BUCKET_CACHE_SIZE = 2
_side_streams = {}
PREFETCH_BUCKETS = os.environ.get("TRS_PREFETCH_BUCKETS", "1") not in ("", "0")
# priority of the side stream the row buckets are built on (HIP: larger number = lower priority)
SIDE_STREAM_PRIORITY = int(os.environ.get("TRS_SIDE_STREAM_PRIORITY", "0"))
# TRS_PREFETCH_EARLY=1: start the bucket build BEFORE the fused lookup launch (beside the HBM-bound lookup instead of beside
# the first MLP GEMM)
PREFETCH_EARLY = os.environ.get("TRS_PREFETCH_EARLY", "0") not in ("", "0")


def side_stream(dev: torch.device, role: str) -> torch.cuda.Stream:
    """The per-device side stream of a ROLE: "buckets" (row-bucket builds, beside the dense forward), "lookup" (the
    lookups of a batch beyond the first, beside it -- and, because autograd runs a node's backward on its forward's stream,
    their bucket walks beside the first one's), "pack" (weight copies into MFMA fragment order, beside the first GEMM of
    a deep branch).  One stream per role: work of one role must never queue behind another's (a weight copy behind a
    bucket build would stall the main stream for the whole build)."""
    side = _side_streams.get((dev, role))
    if side is None:
        side = _side_streams[(dev, role)] = torch.cuda.Stream(device=dev, priority=SIDE_STREAM_PRIORITY)
    return side


def run_on_side(dev: torch.device, role: str, fn):
    """Enqueue ``fn()`` on the role's side stream behind everything the current stream holds so far; returns (result,
    event recorded behind it, the side stream).  The caller makes the consumer's stream wait for the event and tells the
    allocator about tensors that cross (record_stream).  Capture-safe: the side stream forks from the current stream and
    the event wait joins it again."""
    side = side_stream(dev, role)
    main = _abi.current_stream_of(dev)
    side.wait_stream(main)
    torch.cuda.set_stream(side)      # (not ``with torch.cuda.stream``: see prefetch_row_buckets)
    try:
        out = fn()
    finally:
        torch.cuda.set_stream(main)
    ev = torch.cuda.Event()
    ev.record(side)
    return out, ev, side


def _adopt_grads(*grads):
    """A lookup's backward runs on the stream of its forward -- the "lookup" side stream for every lookup of a batch but the
    first -- while the gradients it receives were allocated (and will be freed) under the producer's stream: tell the
    allocator.  No-op until a side lookup has happened."""
    for g in grads:
        if g is not None and g.is_cuda and (g.device, "lookup") in _side_streams:
            cur = _abi.current_stream_of(g.device)
            (g.rename(None) if g.has_names() else g).record_stream(cur)


_offsets_content = {}     # offsets tensor (ptr, version) -> tuple of its values (read back once)


def _offsets_key(offsets):
    """Content key of a per-field offsets vector, so two modules built from the same field sizes (the E=64
    table and the E=1 first-order table of one model) share the row buckets of a batch."""
    if offsets is None:
        return None
    k = (offsets.data_ptr(), offsets._version, offsets.numel())
    v = _offsets_content.get(k)
    if v is None:
        if len(_offsets_content) > 256:
            _offsets_content.clear()
        v = _offsets_content[k] = tuple(offsets.tolist())
    return v


def _bucket_key(idx, offsets, V, check=True):
    # ``check`` is part of the key: a build that skipped out-of-range ids silently (fixed-capacity padding) must not be
    # handed to a caller that expects the index flag to have been raised for them, nor the other way round
    return (idx.data_ptr(), idx._version, tuple(idx.shape), idx.dtype, _offsets_key(offsets), V, bool(check))


def _build_buckets(idx, offsets, V, check: bool = True) -> RowBuckets:
    """``check=False``: ids outside [0, V) are skipped WITHOUT raising the index flag (the padding slots of a
    fixed-capacity exchange carry -1; real ids were range-checked when they were looked up)."""
    B, N = idx.shape
    BN = B * N
    dev = idx.device
    row_start = torch.empty(V + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(BN, 1), dtype=torch.int32, device=dev)
    ws_bytes = size_query("trs_csr_workspace_bytes", V, BN)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    flag = _ErrFlag(dev) if check else None
    call("trs_csr_build", ptr(idx), index_dtype_code(idx), ptr(offsets), B, N, V, ptr(row_start), ptr(perm),
         ptr(ws), ws_bytes, ptr(flag.t if check else None), stream_ptr())
    if check:
        flag.check("row_buckets")
    return RowBuckets(row_start, perm, V, BN, N)


def row_buckets(idx: torch.Tensor, offsets: Optional[torch.Tensor], V: int, check: bool = True) -> RowBuckets:
    """Build (or fetch) the CSR for ``idx`` (B,N).  The cache keeps a reference to the index tensor,
    so its storage cannot be recycled for another batch while the entry is live; in-place edits bump
    ``_version`` and miss."""
    key = _bucket_key(idx, offsets, V, check)
    for k, _, rb in _bucket_cache:
        if k == key:
            rb.wait()
            return rb
    rb = _build_buckets(idx, offsets, V, check)
    # an inline build is shared like a prefetched one (two tables looked up with the same indices, possibly on two
    # streams): it carries its event and stream too
    rb.built_on(_abi.current_stream_of(idx.device))
    _bucket_cache.append((key, idx, rb))
    if len(_bucket_cache) > _BUCKET_CACHE_SIZE:
        _bucket_cache.pop(0)
    return rb


_pending_prefetch: List[tuple] = []
_defer_depth = [0]


class defer_prefetch:
    """Context manager: bucket prefetches requested inside are launched when the block exits.  ``Inputs.forward``
    wraps its loop over the embedding modules with it, so the (atomics-heavy) bucket build starts after ALL
    lookups of the batch have been enqueued and overlaps the dense part of the model instead of the lookups."""

    def __enter__(self):
        _defer_depth[0] += 1
        return self

    def __exit__(self, *exc):
        _defer_depth[0] -= 1
        if _defer_depth[0] == 0:
            reqs = list(_pending_prefetch)
            _pending_prefetch.clear()
            for idx, offsets, V, check in reqs:
                prefetch_row_buckets(idx, offsets, V, check)
        return False


def prefetch_row_buckets(idx: torch.Tensor, offsets: Optional[torch.Tensor], V: int, check: bool = True,
                         now: bool = False) -> None:
    """Start building the CSR of this batch on a side stream at FORWARD time: it depends only on the
    indices, is latency-bound (int32 atomics), and hides behind the forward/backward of the dense part
    of the model; the backward's scatter then just waits on an event."""
    if not PREFETCH_BUCKETS:
        return
    if _defer_depth[0] > 0 and not now:
        _pending_prefetch.append((idx, offsets, V, check))
        return
    key = _bucket_key(idx, offsets, V, check)
    for k, _, _rb in _bucket_cache:
        if k == key:
            return
    dev = idx.device
    side = side_stream(dev, "buckets")
    main = _abi.current_stream_of(dev)
    side.wait_stream(main)
    torch.cuda.set_stream(side)          # not `with torch.cuda.stream(side)`: its constructor and __enter__ each resolve
    try:                                 # the current device through hipGetDeviceCount (~0.1 ms apiece)
        rb = _build_buckets(idx, offsets, V, check)
        rb.built_on(side)
    finally:
        torch.cuda.set_stream(main)
    idx.record_stream(side)
    if offsets is not None:
        offsets.record_stream(side)
    _bucket_cache.append((key, idx, rb))
    if len(_bucket_cache) > _BUCKET_CACHE_SIZE:
        _bucket_cache.pop(0)


_clear_hooks: List = []        # other per-batch caches keyed by index-tensor identity (dist.py's route plans)


def clear_caches():
    _bucket_cache.clear()
    for hook in _clear_hooks:
        hook()


# --------------------------------------------------------------------------------------------
# N2: index staging -- per-field columns -> one (B, N) index matrix
# --------------------------------------------------------------------------------------------
def pack_columns_supported(cols: Sequence[torch.Tensor]) -> bool:
    """One-pass packing applies to >= 2 integer (int64 / int32, one dtype) contiguous HIP tensors of shape (B,) or
    (B, k) with the same B whose total width fits the kernel's tile."""
    if len(cols) < 2:
        return False
    c0 = cols[0]
    if not c0.is_cuda or c0.dtype not in (torch.int64, torch.int32) or c0.dim() not in (1, 2):
        return False
    W = 0
    for c in cols:
        if (not c.is_cuda or c.device != c0.device or c.dtype != c0.dtype or c.dim() not in (1, 2)
                or c.shape[0] != c0.shape[0] or not c.is_contiguous() or c.has_names()):
            return False
        W += 1 if c.dim() == 1 else c.shape[1]
    return 0 < W <= (127 if c0.dtype == torch.int64 else 255)


def pack_columns(cols: Sequence[torch.Tensor], out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``torch.cat([c.unsqueeze(-1) if c.dim() == 1 else c for c in cols], dim=1)`` in one kernel
    (inputs/inputs.py:75-80); ``out_dtype=torch.int32`` narrows on the way (caller guarantees the range)."""
    if not pack_columns_supported(cols):
        raise ValueError("pack_columns: need >= 2 contiguous int64/int32 HIP tensors of shape (B,) or (B,k), same B")
    c0 = cols[0]
    B = c0.shape[0]
    widths = [1 if c.dim() == 1 else int(c.shape[1]) for c in cols]
    out_dtype = out_dtype or c0.dtype
    if out_dtype not in (torch.int64, torch.int32):
        raise TypeError(f"pack_columns: out_dtype {out_dtype}")
    out = torch.empty(B, sum(widths), dtype=out_dtype, device=c0.device)
    n = len(cols)
    srcs = (ctypes.c_void_p * n)(*[c.data_ptr() for c in cols])
    wid = (ctypes.c_int32 * n)(*widths)
    call("trs_pack_columns", srcs, wid, n, index_dtype_code(c0), B, ptr(out),
         _abi.TRS_I64 if out_dtype == torch.int64 else _abi.TRS_I32, stream_ptr())
    return out


def _bcast_cols(g_bcast: Optional[torch.Tensor], E: int) -> int:
    """trs_scatter_rows' g_fm_cols: E for full (B,E) rows, 1 for one value per sample ((B,) or (B,1) with E > 1)"""
    if g_bcast is None:
        return E
    return 1 if (g_bcast.dim() == 1 or g_bcast.shape[-1] == 1) else E


def _fm_grad_operand(g_fm: torch.Tensor) -> torch.Tensor:
    """The FM gradient as trs_scatter_rows wants it: an expanded (B,1) -> (B,E) gradient (what a ``sum`` over E feeds
    back) is passed as its (B,1) column -- the kernel then reads one value per sample instead of a row of E"""
    if g_fm.dim() == 2 and g_fm.shape[1] > 1 and g_fm.stride(1) == 0:
        return g_fm[:, :1].contiguous()
    return g_fm.contiguous()


def scatter_rows(rb: RowBuckets, like_table: torch.Tensor, g_rows: Optional[torch.Tensor] = None,
                 g_bcast: Optional[torch.Tensor] = None, fm_sum: Optional[torch.Tensor] = None,
                 padding_row: int = -1, g_rows_batch_stride: int = 0) -> torch.Tensor:
    """Dense gradient of a (V,E) table: see trs_scatter_rows in include/trs_abi.h."""
    V, E = like_table.shape
    grad = torch.empty(V, E, dtype=like_table.dtype, device=like_table.device)
    ws_bytes = size_query("trs_scatter_workspace_bytes", rb.BN, rb.N, E, value_dtype_code(like_table))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=like_table.device)
    call("trs_scatter_rows", ptr(g_rows), g_rows_batch_stride, ptr(g_bcast), _bcast_cols(g_bcast, E), ptr(fm_sum),
         ptr(like_table if fm_sum is not None else None), ptr(rb.row_start), ptr(rb.perm), rb.BN, V, E, rb.N,
         value_dtype_code(like_table), padding_row, ptr(grad), ptr(ws), ws_bytes, stream_ptr())
    return grad


def scatter_rows_first(rb: RowBuckets, like_table: torch.Tensor, like_first: torch.Tensor, g_first: torch.Tensor,
                       g_rows: Optional[torch.Tensor] = None, g_bcast: Optional[torch.Tensor] = None,
                       fm_sum: Optional[torch.Tensor] = None, padding_row: int = -1
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense gradients of a (V,E) table and of its first-order companion (V,1) in one bucket walk: ``g_first`` holds
    one value per lookup (B,N[,1]).  See trs_scatter_rows_first in include/trs_abi.h."""
    V, E = like_table.shape
    grad = torch.empty_like(like_table)
    gfirst = torch.empty_like(like_first)
    ws_bytes = size_query("trs_scatter_workspace_bytes", rb.BN, rb.N, E, value_dtype_code(like_table))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=like_table.device)
    call("trs_scatter_rows_first", ptr(g_rows), 0, ptr(g_bcast), _bcast_cols(g_bcast, E), ptr(fm_sum),
         ptr(like_table if fm_sum is not None else None), ptr(rb.row_start), ptr(rb.perm), rb.BN, V, E, rb.N,
         value_dtype_code(like_table), padding_row, ptr(grad), ptr(g_first), ptr(gfirst), ptr(ws), ws_bytes, stream_ptr())
    return grad, gfirst


def scatter_rows_update(rb: RowBuckets, table: torch.Tensor, opt, g_rows: Optional[torch.Tensor] = None,
                        g_bcast: Optional[torch.Tensor] = None, fm_sum: Optional[torch.Tensor] = None,
                        padding_row: int = -1, key=None) -> None:
    """Fused sparse optimizer step: the bucketed gradient of every looked-up row is applied to ``table`` in
    place (see trs_scatter_rows_update); no gradient tensor is produced."""
    V, E = table.shape
    if not table.is_contiguous():
        raise ValueError("fused optimizer needs a contiguous table")
    ws_bytes = size_query("trs_scatter_workspace_bytes", rb.BN, rb.N, E, value_dtype_code(table))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=table.device)
    if opt.kind == 3:
        if torch.cuda.is_current_stream_capturing():
            # the bias-corrected step size lr*sqrt(1-b2^t)/(1-b1^t) is computed on the host per step and passed by
            # value: a captured launch would replay the capture-time t for ever (~0.18*lr with default betas)
            raise RuntimeError("torecsys_amd: FusedSparseAdam cannot be captured into a hipGraph (its per-step bias "
                               "correction is a host-side scalar); use FusedSparseSGD / FusedSparseAdagrad under "
                               "GraphedStep, or run the Adam step eagerly")
        m1, m2 = opt.state_for(table, key)
        call("trs_scatter_rows_update_adam", ptr(g_rows), 0, ptr(g_bcast), _bcast_cols(g_bcast, E), ptr(fm_sum), ptr(table),
             ptr(rb.row_start),
             ptr(rb.perm), rb.BN, V, E, rb.N, value_dtype_code(table), padding_row, float(opt.next_step_size(table, key)),
             float(opt.beta1), float(opt.beta2), float(opt.eps), ptr(m1), ptr(m2), ptr(ws), ws_bytes, stream_ptr())
        return
    state = opt.state_for(table, key)
    call("trs_scatter_rows_update", ptr(g_rows), 0, ptr(g_bcast), _bcast_cols(g_bcast, E), ptr(fm_sum), ptr(table),
         ptr(rb.row_start),
         ptr(rb.perm), rb.BN, V, E, rb.N, value_dtype_code(table), padding_row, opt.kind, float(opt.lr), float(opt.eps),
         ptr(state), ptr(ws), ws_bytes, stream_ptr())


def scatter_rows_update_mapped(rb: RowBuckets, table: torch.Tensor, opt, g_rows: torch.Tensor, row_map: torch.Tensor,
                               key=None):
    """Fused sparse optimizer step where the bucketed rows are a compact list of distinct table rows:
    ``rb`` buckets the K rows of ``g_rows`` by compact row u (rb.V = U), ``row_map[u]`` (int32, distinct) is the table
    row that u updates.  See trs_scatter_rows_update_mapped."""
    V, E = table.shape
    if not table.is_contiguous():
        raise ValueError("fused optimizer needs a contiguous table")
    if row_map.dtype != torch.int32 or row_map.numel() != rb.V:
        raise ValueError("row_map must be int32 with one entry per bucketed row")
    ws_bytes = size_query("trs_scatter_workspace_bytes", rb.BN, 1, E, value_dtype_code(table))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=table.device)
    lr, st1, st2, b1, b2 = float(opt.lr), None, None, 0.0, 0.0
    if opt.kind == 3:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("torecsys_amd: FusedSparseAdam cannot be captured into a hipGraph")
        st1, st2 = opt.state_for(table, key)
        lr, b1, b2 = float(opt.next_step_size(table, key)), float(opt.beta1), float(opt.beta2)
    elif opt.kind == 2:
        st1 = opt.state_for(table, key)
    call("trs_scatter_rows_update_mapped", ptr(g_rows.contiguous()), ptr(table), ptr(row_map), ptr(rb.row_start),
         ptr(rb.perm), rb.BN, rb.V, V, E, value_dtype_code(table), opt.kind, lr, float(opt.eps), b1, b2, ptr(st1), ptr(st2),
         ptr(ws), ws_bytes, stream_ptr())


def _apply_or_grad(rb, weight, opt, key=None, **kw):
    """dense gradient (default) or in-place fused optimizer step (returns None).  ``key``: the object the optimizer
    state is filed under -- the parameter itself when ``weight`` is a reshaped view of it (a view is a new Python object
    on every call: keyed by the view, the state would be re-created, and leaked, every step)."""
    if opt is None:
        return scatter_rows(rb, weight, **kw)
    with torch.no_grad():
        scatter_rows_update(rb, weight.data, opt, key=weight if key is None else key, **kw)
    return None


# --------------------------------------------------------------------------------------------
# K1: gather
# --------------------------------------------------------------------------------------------
class _GatherRows(Function):
    @staticmethod
    def forward(ctx, weight, idx, offsets, padding_idx, opt=None):
        require_device(weight, idx, offsets)
        B, N = idx.shape
        V, E = weight.shape
        w = weight.contiguous()
        out = torch.empty(B, N, E, dtype=w.dtype, device=w.device)
        flag = _ErrFlag(w.device)
        call("trs_gather_rows", ptr(w), V, E, value_dtype_code(w), ptr(idx), index_dtype_code(idx), ptr(offsets),
             B, N, ptr(out), ptr(flag.t), stream_ptr())
        flag.check("gather_rows")
        if ctx.needs_input_grad[0] and E * w.element_size() >= 16:      # False under torch.no_grad(): no backward, no buckets
            # (narrow tables -- the E=1 first-order weights -- leave the prefetch to the wide table that shares
            # their indices, so the bucket build overlaps the dense part of the model, not the lookups)
            prefetch_row_buckets(idx, offsets, V)
        ctx.save_for_backward(idx, offsets, weight)
        ctx.padding_idx = -1 if padding_idx is None else int(padding_idx)
        ctx.opt = opt
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        idx, offsets, weight = ctx.saved_tensors
        _adopt_grads(g)
        rb = row_buckets(idx, offsets, weight.shape[0])
        if g.dim() == 3 and g.shape[1] > 1 and g.stride(1) == 0:
            # the same gradient row for every field of a sample (the backward of a sum over the fields, e.g. the models'
            # first-order term): read as one (B,E) row per sample instead of materialising the expanded (B,N,E) block
            grad = _apply_or_grad(rb, weight, ctx.opt, g_bcast=g[:, 0, :].contiguous(), padding_row=ctx.padding_idx)
        else:
            grad = _apply_or_grad(rb, weight, ctx.opt, g_rows=g.contiguous(), padding_row=ctx.padding_idx)
        return grad, None, None, None, None


def gather_rows(weight: torch.Tensor, idx: torch.Tensor, offsets: Optional[torch.Tensor] = None,
                padding_idx: Optional[int] = None, opt=None) -> torch.Tensor:
    """out[b,n,:] = weight[idx[b,n] + offsets[n], :]; dense-gradient backward."""
    idx = _as_index(idx)
    if idx.dim() == 1:
        idx = idx.unsqueeze(-1)
    if idx.dim() != 2:
        raise ValueError(f"indices must be (B, N), got shape {tuple(idx.shape)}")
    if padding_idx is not None and padding_idx < 0:
        padding_idx = weight.shape[0] + padding_idx
    return _GatherRows.apply(weight, idx, offsets, padding_idx, opt)


# --------------------------------------------------------------------------------------------
# K1+K2(+K8): fused lookup + FM (+ first-order sum)
# --------------------------------------------------------------------------------------------
class _EmbedFM(Function):
    """``fields=True``: the first-order output is the per-field tensor (B,N,1) the reference's models consume (one value
    per lookup, trs_embed_fm_fields) instead of its sum (B,1); the backward then takes both tables' dense gradients from
    one bucket walk (trs_scatter_rows_first)."""

    @staticmethod
    def forward(ctx, weight, idx, offsets, first_weight, want_emb, opt=None, fields=False, padding_idx=None):
        require_device(weight, idx, offsets, first_weight)
        B, N = idx.shape
        V, E = weight.shape
        w = weight.contiguous()
        dev = w.device
        emb = torch.empty(B, N, E, dtype=w.dtype, device=dev) if want_emb else None
        fm = torch.empty(B, E, dtype=w.dtype, device=dev)
        fm_sum = torch.empty(B, E, dtype=torch.float32, device=dev)
        first = None
        fw = None
        if first_weight is not None:
            if first_weight.dtype != w.dtype or first_weight.numel() != V:
                raise ValueError("first-order table must be (V,1) with the embedding table's dtype")
            fw = first_weight.contiguous()
            first = torch.empty((B, N, 1) if fields else (B, 1), dtype=w.dtype, device=dev)
        elif fields:
            raise ValueError("fields=True needs the first-order table")
        flag = _ErrFlag(dev)
        if PREFETCH_EARLY and (ctx.needs_input_grad[0] or ctx.needs_input_grad[3]):
            prefetch_row_buckets(idx, offsets, V, now=True)
        call("trs_embed_fm_fields" if fields else "trs_embed_fm", ptr(w), V, E, value_dtype_code(w), ptr(idx),
             index_dtype_code(idx), ptr(offsets), B, N, ptr(emb), ptr(fm), ptr(fm_sum), ptr(fw), ptr(first), ptr(flag.t),
             stream_ptr())
        flag.check("embed_fm")
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
            prefetch_row_buckets(idx, offsets, V)
        ctx.save_for_backward(idx, offsets, weight, first_weight, fm_sum)
        ctx.want_emb = want_emb
        ctx.opt = opt
        ctx.fields = bool(fields)
        ctx.padding_idx = -1 if padding_idx is None else int(padding_idx)
        ctx.set_materialize_grads(False)   # unused outputs arrive as None, not as zero blocks
        outs = (emb if want_emb else fm.new_empty(0), fm, first if first is not None else fm.new_empty(0))
        if not want_emb:
            ctx.mark_non_differentiable(outs[0])
        if first is None:
            ctx.mark_non_differentiable(outs[2])
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_emb, g_fm, g_first):
        idx, offsets, weight, first_weight, fm_sum = ctx.saved_tensors
        _adopt_grads(g_emb, g_fm, g_first)
        V, E = weight.shape
        rb = row_buckets(idx, offsets, V)
        gw = gfw = None
        has_emb = ctx.want_emb and g_emb is not None
        has_fm = g_fm is not None
        pad = ctx.padding_idx      # nn.Embedding(padding_idx=): that row of the E-wide table receives no gradient
        if (ctx.fields and ctx.opt is None and g_first is not None and (has_emb or has_fm) and pad < 0
                and ctx.needs_input_grad[0] and ctx.needs_input_grad[3] and (E * weight.element_size()) % 16 == 0):
            gw, gfw = scatter_rows_first(rb, weight, first_weight, g_first.contiguous(),
                                         g_rows=g_emb.contiguous() if has_emb else None,
                                         g_bcast=_fm_grad_operand(g_fm) if has_fm else None,
                                         fm_sum=fm_sum if has_fm else None)
            return gw, None, None, gfw, None, None, None, None
        if ctx.needs_input_grad[0]:
            if has_emb or has_fm:
                gw = _apply_or_grad(rb, weight, ctx.opt, g_rows=g_emb.contiguous() if has_emb else None,
                                    g_bcast=_fm_grad_operand(g_fm) if has_fm else None, fm_sum=fm_sum if has_fm else None,
                                    padding_row=pad)
            elif ctx.opt is None:
                gw = torch.zeros_like(weight)
        if first_weight is not None and ctx.needs_input_grad[3]:
            if g_first is not None:
                if ctx.fields:       # one gradient value per lookup: the E = 1 table's own bucketed scatter
                    gfw = _apply_or_grad(rb, first_weight.reshape(V, 1), ctx.opt, key=first_weight,
                                         g_rows=g_first.contiguous())
                else:
                    gfw = _apply_or_grad(rb, first_weight.reshape(V, 1), ctx.opt, key=first_weight,
                                         g_bcast=g_first.contiguous().reshape(-1, 1))
                gfw = None if gfw is None else gfw.reshape(first_weight.shape)
            elif ctx.opt is None:
                gfw = torch.zeros_like(first_weight)
        return gw, None, None, gfw, None, None, None, None


def embed_fm(weight: torch.Tensor, idx: torch.Tensor, offsets: Optional[torch.Tensor] = None,
             first_weight: Optional[torch.Tensor] = None, want_emb: bool = True, opt=None,
             padding_idx: Optional[int] = None
             ) -> Tuple[Optional[torch.Tensor], torch.Tensor, Optional[torch.Tensor]]:
    """One pass over the looked-up rows: (emb (B,N,E) | None, fm (B,E), first (B,1) | None).  ``padding_idx``: the
    table row nn.Embedding(padding_idx=) keeps gradient-free (multi_indices_emb.py:48 forwards it); the forward reads
    that row like any other, as F.embedding does."""
    idx = _as_index(idx)
    if idx.dim() != 2:
        raise ValueError(f"indices must be (B, N), got shape {tuple(idx.shape)}")
    if padding_idx is not None and padding_idx < 0:
        padding_idx = weight.shape[0] + padding_idx
    emb, fm, first = _EmbedFM.apply(weight, idx, offsets, first_weight, want_emb, opt, False, padding_idx)
    return (emb if want_emb else None), fm, (first if first_weight is not None else None)


def embed_fm_fields(weight: torch.Tensor, first_weight: torch.Tensor, idx: torch.Tensor,
                    offsets: Optional[torch.Tensor] = None, opt=None
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The lookup of an embedding table, the FM term of the looked-up rows and the per-field lookup of the first-order
    table that shares the indices, in one pass: (emb (B,N,E), fm (B,E), first (B,N,1))."""
    idx = _as_index(idx)
    if idx.dim() != 2:
        raise ValueError(f"indices must be (B, N), got shape {tuple(idx.shape)}")
    return _EmbedFM.apply(weight, idx, offsets, first_weight, True, opt, True)


# --------------------------------------------------------------------------------------------
# K2: FM on a materialised block
# --------------------------------------------------------------------------------------------
class _FMLayer(Function):
    @staticmethod
    def forward(ctx, x):
        require_device(x)
        x = x.contiguous()
        B, N, E = x.shape
        fm = torch.empty(B, E, dtype=x.dtype, device=x.device)
        fm_sum = torch.empty(B, E, dtype=torch.float32, device=x.device)
        call("trs_fm_fwd", ptr(x), B, N, E, value_dtype_code(x), ptr(fm), ptr(fm_sum), stream_ptr())
        ctx.save_for_backward(x, fm_sum)
        return fm

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, fm_sum = ctx.saved_tensors
        B, N, E = x.shape
        dx = torch.empty_like(x)
        call("trs_fm_bwd", ptr(x), ptr(g.contiguous()), ptr(fm_sum), B, N, E, value_dtype_code(x), ptr(dx),
             stream_ptr())
        return dx


def fm_layer(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 3:
        raise ValueError(f"FM input must be (B, N, E), got {tuple(x.shape)}")
    return _FMLayer.apply(x)


# --------------------------------------------------------------------------------------------
# I3: field-aware gather  (B,N) -> (B,N*N,E)
# --------------------------------------------------------------------------------------------
_ptr_table_cache = {}


def _pointer_table(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    """Device array of base pointers (cached per pointer tuple: a blocking 8*N-byte upload otherwise)."""
    key = tuple(t.data_ptr() for t in tensors)
    tab = _ptr_table_cache.get(key)
    if tab is None:
        if len(_ptr_table_cache) > 64:
            _ptr_table_cache.clear()
        # a pageable host -> device copy with non_blocking=False: PyTorch synchronises the copy's stream before returning,
        # so the entry is complete for every stream that reads it afterwards (lookups run on several streams)
        tab = torch.tensor(key, dtype=torch.int64).to(tensors[0].device, non_blocking=False)
        _ptr_table_cache[key] = tab
    return tab


class _FAGather(Function):
    @staticmethod
    def forward(ctx, idx, offsets, *weights):
        require_device(idx, offsets, *weights)
        B, N = idx.shape
        if len(weights) != N:
            raise ValueError(f"need {N} tables, got {len(weights)}")
        V, E = weights[0].shape
        ws = [w.contiguous() for w in weights]
        for w in ws:
            if w.shape != (V, E) or w.dtype != ws[0].dtype:
                raise ValueError("field-aware tables must share shape and dtype")
        out = torch.empty(B, N * N, E, dtype=ws[0].dtype, device=ws[0].device)
        flag = _ErrFlag(out.device)
        call("trs_fa_gather_rows", ptr(_pointer_table(ws)), V, E, value_dtype_code(ws[0]), ptr(idx),
             index_dtype_code(idx), ptr(offsets), B, N, ptr(out), ptr(flag.t), stream_ptr())
        flag.check("fa_gather_rows")
        if any(ctx.needs_input_grad[2:]):
            prefetch_row_buckets(idx, offsets, V)
        ctx.save_for_backward(idx, offsets, *weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        idx, offsets, *weights = ctx.saved_tensors
        _adopt_grads(g)
        B, N = idx.shape
        V, E = weights[0].shape
        g = g.contiguous()
        rb = row_buckets(idx, offsets, V)
        grads = []
        for i, w in enumerate(weights):
            if ctx.needs_input_grad[2 + i]:
                grads.append(scatter_rows(rb, w, g_rows=g[:, i * N:(i + 1) * N], g_rows_batch_stride=N * N))
            else:
                grads.append(None)
        return (None, None, *grads)


_table_rows_cache = {}


def _tables_meta(weights: Sequence[torch.Tensor]):
    """(row counts, cumulative row offsets) of N tables as device int64 tensors, cached per shape tuple and device"""
    key = (tuple(int(w.shape[0]) for w in weights), weights[0].device)
    m = _table_rows_cache.get(key)
    if m is None:
        if len(_table_rows_cache) > 64:
            _table_rows_cache.clear()
        rows = torch.tensor(key[0], dtype=torch.int64)
        off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(rows, 0)[:-1]])
        m = _table_rows_cache[key] = (rows.to(key[1], non_blocking=False), off.to(key[1], non_blocking=False))  # blocking: see _pointer_table
    return m


class _GatherRowsTables(Function):
    """out[b,n,:] = weights[n][idx[b,n],:] for N SEPARATE tables of one width (a StackedInput of SingleIndexEmbeddings,
    stacked_inp.py:94-134) in one launch (trs_gather_rows_tables).  Backward: one row-bucket build over the concatenated
    row space and one bucket walk into a (sum V_n, E) gradient; every table's gradient is its slice of that tensor."""

    @staticmethod
    def forward(ctx, idx, *weights):
        require_device(idx, *weights)
        B, N = idx.shape
        E = weights[0].shape[1]
        if len(weights) != N:
            raise ValueError(f"gather_rows_tables: {len(weights)} tables for {N} index columns")
        ws = [w.contiguous() for w in weights]
        for w in ws:
            if w.dim() != 2 or w.shape[1] != E or w.dtype != ws[0].dtype:
                raise ValueError("gather_rows_tables: the tables must share embed size and dtype")
            if (E * w.element_size()) % 16 == 0 and w.data_ptr() % 16 != 0:
                raise ValueError("gather_rows_tables: every table must be 16-byte aligned")
        rows, offsets = _tables_meta(ws)
        out = torch.empty(B, N, E, dtype=ws[0].dtype, device=ws[0].device)
        flag = _ErrFlag(out.device)
        call("trs_gather_rows_tables", ptr(_pointer_table(ws)), ptr(rows), E, value_dtype_code(ws[0]), ptr(idx),
             index_dtype_code(idx), B, N, ptr(out), ptr(flag.t), stream_ptr())
        flag.check("gather_rows_tables")
        if any(ctx.needs_input_grad[1:]):
            prefetch_row_buckets(idx, offsets, int(sum(w.shape[0] for w in ws)))
        ctx.save_for_backward(idx, offsets)
        ctx.shapes = [tuple(w.shape) for w in ws]
        ctx.like = ws[0]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        idx, offsets = ctx.saved_tensors
        _adopt_grads(g)
        Vs = [s[0] for s in ctx.shapes]
        Vtot, E = sum(Vs), ctx.shapes[0][1]
        rb = row_buckets(idx, offsets, Vtot)
        gcat = scatter_rows(rb, _ShapeOnly(Vtot, E, ctx.like), g_rows=g.contiguous())
        grads, o = [], 0
        for i, v in enumerate(Vs):
            grads.append(gcat[o:o + v] if ctx.needs_input_grad[1 + i] else None)
            o += v
        return (None, *grads)


class _ShapeOnly:
    """what scatter_rows needs of its ``like_table`` when no (V,E) table exists: shape, dtype, device"""

    def __init__(self, V, E, like):
        self.shape, self.dtype, self.device, self._like = (V, E), like.dtype, like.device, like

    def element_size(self):
        return self._like.element_size()


def gather_rows_tables(weights: Sequence[torch.Tensor], idx: torch.Tensor) -> torch.Tensor:
    """(B,N) raw per-table indices -> (B,N,E) from N separate (V_n,E) tables; dense per-table gradients"""
    idx = _as_index(idx)
    if idx.dim() == 1:
        idx = idx.unsqueeze(-1)
    if idx.dim() != 2:
        raise ValueError(f"indices must be (B, N), got shape {tuple(idx.shape)}")
    return _GatherRowsTables.apply(idx, *weights)


def fa_gather_rows(weights: Sequence[torch.Tensor], idx: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    idx = _as_index(idx)
    return _FAGather.apply(idx, offsets, *weights)


# --------------------------------------------------------------------------------------------
# K7: inner-product network
# --------------------------------------------------------------------------------------------
class _PairDot(Function):
    @staticmethod
    def forward(ctx, x):
        require_device(x)
        x = x.contiguous()
        B, N, E = x.shape
        out = torch.empty(B, N * (N - 1) // 2, dtype=x.dtype, device=x.device)
        call("trs_pair_dot_fwd", ptr(x), B, N, E, value_dtype_code(x), ptr(out), stream_ptr())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, N, E = x.shape
        dx = torch.empty_like(x)
        call("trs_pair_dot_bwd", ptr(x), ptr(g.contiguous()), B, N, E, value_dtype_code(x), ptr(dx), stream_ptr())
        return dx


def pair_dot(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 3:
        raise ValueError(f"inner-product input must be (B, N, E), got {tuple(x.shape)}")
    return _PairDot.apply(x)


def embed_ipn_supported(weight: torch.Tensor, N: int) -> bool:
    """the lookup + inner-product kernel covers bf16 tables whose rows the matrix-core pair kernel covers"""
    return (weight.is_cuda and weight.dtype == torch.bfloat16 and weight.dim() == 2 and weight.shape[1] in (32, 64, 128)
            and 2 <= N <= 64)


class _EmbedIPN(Function):
    """K7 fused with K1 (trs_embed_pair_dot): (emb (B,N,E), ipn (B,NC2)) in one pass over the table rows.  The block is
    written only when someone needs it: the caller (``want_emb``) or the backward (it is the ``x`` of trs_pair_dot_bwd);
    at inference with ``want_emb=False`` the (B,N,E) block never exists."""

    @staticmethod
    def forward(ctx, weight, idx, offsets, want_emb, opt=None):
        require_device(weight, idx, offsets)
        B, N = idx.shape
        V, E = weight.shape
        w = weight.contiguous()
        need_block = want_emb or ctx.needs_input_grad[0]
        emb = torch.empty(B, N, E, dtype=w.dtype, device=w.device) if need_block else None
        out = torch.empty(B, N * (N - 1) // 2, dtype=w.dtype, device=w.device)
        flag = _ErrFlag(w.device)
        call("trs_embed_pair_dot", ptr(w), V, E, value_dtype_code(w), ptr(idx), index_dtype_code(idx), ptr(offsets), B, N,
             ptr(emb), ptr(out), ptr(flag.t), stream_ptr())
        flag.check("embed_pair_dot")
        if ctx.needs_input_grad[0]:
            prefetch_row_buckets(idx, offsets, V)
        ctx.save_for_backward(idx, offsets, weight, emb if ctx.needs_input_grad[0] else None)
        ctx.want_emb = want_emb
        ctx.opt = opt
        ctx.set_materialize_grads(False)
        if not want_emb:
            blk = out.new_empty(0)
            ctx.mark_non_differentiable(blk)
            return blk, out
        return emb, out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_emb, g_out):
        idx, offsets, weight, emb = ctx.saved_tensors
        _adopt_grads(g_emb, g_out)
        V, E = weight.shape
        B, N = idx.shape
        g_rows = None
        if g_out is not None:
            g_rows = torch.empty_like(emb)
            call("trs_pair_dot_bwd", ptr(emb), ptr(g_out.contiguous()), B, N, E, value_dtype_code(emb), ptr(g_rows),
                 stream_ptr())
        if ctx.want_emb and g_emb is not None:
            g_rows = g_emb.contiguous() if g_rows is None else g_rows.add_(g_emb)
        if g_rows is None:
            return (torch.zeros_like(weight) if ctx.opt is None else None), None, None, None, None
        rb = row_buckets(idx, offsets, V)
        return _apply_or_grad(rb, weight, ctx.opt, g_rows=g_rows), None, None, None, None


def embed_ipn(weight: torch.Tensor, idx: torch.Tensor, offsets: Optional[torch.Tensor] = None, want_emb: bool = True,
              opt=None) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """Lookup + inner-product network in one kernel: (emb (B,N,E) | None, ipn (B, N(N-1)/2))."""
    idx = _as_index(idx)
    if idx.dim() != 2:
        raise ValueError(f"indices must be (B, N), got shape {tuple(idx.shape)}")
    emb, out = _EmbedIPN.apply(weight, idx, offsets, want_emb, opt)
    return (emb if want_emb else None), out


# --------------------------------------------------------------------------------------------
# F4 glue: BatchNorm1d + ReLU + direct/hidden split + pooled sum on channels-last CIN activations
# --------------------------------------------------------------------------------------------
def cin_glue_supported(yT: torch.Tensor, D: int, Hs: int) -> bool:
    C = yT.shape[-1]
    vpr = C // 8
    return (yT.is_cuda and yT.dtype == torch.bfloat16 and yT.dim() == 3 and yT.is_contiguous() and C % 8 == 0
            and 1 <= vpr <= 256 and 256 % vpr == 0 and D % 8 == 0 and Hs % 8 == 0)


class _CINGlue(Function):
    """(hidden (B,E,C-Hs), pooled (B,D)) from the contraction output y (B,E,C): z = relu(bn(y)), hidden = z[..., Hs:],
    pooled = z[..., :D].sum(1).  BatchNorm1d semantics (batch statistics + running-stat update in training, running
    statistics in eval, affine optional) are reproduced here; see trs_cin_glue_*."""

    @staticmethod
    def forward(ctx, yT, gamma, beta, running_mean, running_var, training, momentum, eps, D, Hs):
        require_device(yT)
        B, E, C = yT.shape
        dev = yT.device
        R = B * E
        use_batch = bool(training) or running_mean is None
        if gamma is None and running_mean is None and not training and momentum is None and eps is None:
            use_batch = False                      # no BatchNorm module at all: identity statistics
        if eps is None:
            mean = torch.zeros(C, dtype=torch.float32, device=dev)
            invstd = torch.ones(C, dtype=torch.float32, device=dev)
            use_batch = False
        elif use_batch:
            nblk = size_query("trs_cin_glue_blocks", B)
            part = torch.empty(nblk, 2, C, dtype=torch.float32, device=dev)
            call("trs_cin_glue_stats", ptr(yT), B, E, C, value_dtype_code(yT), ptr(part), stream_ptr())
            tot = part.double().sum(0)
            mean64 = tot[0] / R
            var64 = (tot[1] / R - mean64 * mean64).clamp_min_(0.0)
            mean = mean64.float()
            invstd = (var64 + eps).rsqrt().float()
            if training and running_mean is not None:
                with torch.no_grad():
                    m = float(momentum)
                    running_mean.mul_(1.0 - m).add_(mean64.to(running_mean.dtype), alpha=m)
                    unbiased = var64 * (R / max(R - 1, 1))
                    running_var.mul_(1.0 - m).add_(unbiased.to(running_var.dtype), alpha=m)
        else:
            mean = running_mean.float()
            invstd = (running_var.double() + eps).rsqrt().float()
        g32 = gamma.float() if gamma is not None else torch.ones(C, dtype=torch.float32, device=dev)
        b32 = beta.float() if beta is not None else torch.zeros(C, dtype=torch.float32, device=dev)
        scale = (g32 * invstd).contiguous()
        shift = (b32 - mean * scale).contiguous()
        hidden = torch.empty(B, E, C - Hs, dtype=yT.dtype, device=dev)
        pooled = torch.empty(B, D, dtype=yT.dtype, device=dev)
        cf = C > Hs and bool(size_query("trs_cin_glue_cf_supported", E, C))
        if cf:
            # also emit the hidden half channels-first (B, C-Hs, E): the next layer's weight-gradient kernel reads
            # that layout (otherwise one transposing copy per layer and step); it rides on the returned tensor
            hidden_cf = torch.empty(B, C - Hs, E, dtype=yT.dtype, device=dev)
            call("trs_cin_glue_fwd_cf", ptr(yT), ptr(scale), ptr(shift), B, E, C, D, Hs, value_dtype_code(yT),
                 ptr(hidden), ptr(hidden_cf), ptr(pooled), stream_ptr())
        else:
            hidden_cf = yT.new_empty(0)
            call("trs_cin_glue_fwd", ptr(yT), ptr(scale), ptr(shift), B, E, C, D, Hs, value_dtype_code(yT), ptr(hidden),
                 ptr(pooled), stream_ptr())
        ctx.mark_non_differentiable(hidden_cf)
        ctx.save_for_backward(yT, scale, shift, mean.contiguous(), invstd.contiguous())
        ctx.meta = (use_batch, D, Hs, gamma is not None, beta is not None,
                    None if gamma is None else gamma.dtype)
        ctx.set_materialize_grads(False)
        return hidden, pooled, hidden_cf

    @staticmethod
    @once_differentiable
    def backward(ctx, g_hidden, g_pooled, _g_cf=None):
        yT, scale, shift, mean, invstd = ctx.saved_tensors
        use_batch, D, Hs, has_gamma, has_beta, pdtype = ctx.meta
        B, E, C = yT.shape
        dev = yT.device
        R = B * E
        gh = None if g_hidden is None else g_hidden.contiguous()
        gp = None if g_pooled is None else g_pooled.contiguous()
        nblk = size_query("trs_cin_glue_blocks", B)
        part = torch.empty(nblk, 2, C, dtype=torch.float32, device=dev)
        call("trs_cin_glue_bwd_reduce", ptr(yT), ptr(gh), ptr(gp), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), B, E, C,
             D, Hs, value_dtype_code(yT), ptr(part), stream_ptr())
        tot = part.double().sum(0)
        dbeta, dgamma = tot[0], tot[1]
        if use_batch:
            c1 = (dbeta / R).float().contiguous()
            c2 = (dgamma / R).float().contiguous()
        else:
            c1 = torch.zeros(C, dtype=torch.float32, device=dev)
            c2 = c1
        gy = torch.empty_like(yT)
        if size_query("trs_cin_glue_cf_supported", E, C):
            gy_cf = torch.empty(B, C, E, dtype=yT.dtype, device=dev)      # (B,C,E) copy for trs_cin_dw, see forward
            csum = torch.empty(nblk, C, dtype=torch.float32, device=dev)   # column sums of gy: the conv-bias gradient
            call("trs_cin_glue_bwd_apply_cf", ptr(yT), ptr(gh), ptr(gp), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 ptr(c1), ptr(c2), B, E, C, D, Hs, value_dtype_code(yT), ptr(gy), ptr(gy_cf), ptr(csum), stream_ptr())
            gy._trs_cf = gy_cf
            gy._trs_colsum = (csum, gy._version)
        else:
            call("trs_cin_glue_bwd_apply", ptr(yT), ptr(gh), ptr(gp), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                 ptr(c1), ptr(c2), B, E, C, D, Hs, value_dtype_code(yT), ptr(gy), stream_ptr())
        ggamma = dgamma.to(pdtype) if has_gamma and ctx.needs_input_grad[1] else None
        gbeta = dbeta.to(pdtype) if has_beta and ctx.needs_input_grad[2] else None
        return gy, ggamma, gbeta, None, None, None, None, None, None, None


def cin_glue(yT, bn: Optional[torch.nn.Module], D: int, Hs: int):
    """BatchNorm1d ``bn`` (or None) + ReLU + split + pooled sum on yT (B,E,C) bf16 -> (hidden (B,E,C-Hs), pooled (B,D))."""
    if bn is None:
        return _glue_out(_CINGlue.apply(yT, None, None, None, None, False, None, None, D, Hs))
    momentum = bn.momentum
    if bn.training and bn.track_running_stats:
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
            if momentum is None:                              # cumulative moving average
                momentum = 1.0 / float(bn.num_batches_tracked)
    use_running = (not bn.training) and bn.track_running_stats
    return _glue_out(_CINGlue.apply(yT, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                    bn.running_var if bn.track_running_stats else None,
                                    bn.training or not use_running, momentum if momentum is not None else 0.0, bn.eps,
                                    D, Hs))


def _glue_out(res):
    """(hidden, pooled, hidden_cf) -> (hidden, pooled); the channels-first copy rides on ``hidden`` as ``_trs_cf``"""
    hidden, pooled, hidden_cf = res
    if hidden_cf.numel() > 0:
        hidden._trs_cf = hidden_cf
    return hidden, pooled


# --------------------------------------------------------------------------------------------
# N3: outer-product network / bilinear interaction on the pair pattern
# --------------------------------------------------------------------------------------------
def _pair_start(i: int, N: int) -> int:
    return i * (2 * N - i - 1) // 2


class _OPNVec(Function):
    """out[b,p] = sum_e x_i x_j kern[p,e]  ('vec': kern (P,E); 'num': kern (P,))."""

    @staticmethod
    def forward(ctx, x, kern, is_num):
        require_device(x, kern)
        x = x.contiguous()
        kern = kern.contiguous().to(x.dtype)
        B, N, E = x.shape
        out = torch.empty(B, N * (N - 1) // 2, dtype=x.dtype, device=x.device)
        call("trs_opn_vec_fwd", ptr(x), ptr(kern), int(is_num), B, N, E, value_dtype_code(x), ptr(out), stream_ptr())
        ctx.save_for_backward(x, kern)
        ctx.is_num = bool(is_num)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, kern = ctx.saved_tensors
        B, N, E = x.shape
        P = N * (N - 1) // 2
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gk = torch.zeros(P, E, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        ws_bytes = size_query("trs_opn_vec_bwd_workspace_bytes", B, N, E)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        call("trs_opn_vec_bwd", ptr(g.contiguous()), ptr(x), ptr(kern), int(ctx.is_num), B, N, E, value_dtype_code(x),
             ptr(gx), ptr(gk), ptr(ws), ws_bytes, stream_ptr())
        if gk is not None:
            gk = (gk.sum(-1) if ctx.is_num else gk).to(kern.dtype)
        return gx, gk, None


def opn_vec(x: torch.Tensor, kern: torch.Tensor, is_num: bool) -> torch.Tensor:
    if x.dim() != 3:
        raise ValueError(f"outer-product input must be (B, N, E), got {tuple(x.shape)}")
    return _OPNVec.apply(x, kern, is_num)


def _pair_bias_grad(g: torch.Tensor, per_pair: bool) -> torch.Tensor:
    """sum of a (B, P, E) gradient over b (per-pair bias) or over (b, p) (shared bias), fp32 accumulation."""
    B, P, E = g.shape
    if not per_pair and cin_glue_supported(g, 0, 0):
        # column sums of the (B*P, E) matrix: one bf16 read with fp32 partials per workgroup (trs_cin_glue_stats)
        # instead of a cast to fp32 plus a two-stage ATen reduction (three passes over a multi-GB tensor)
        nblk = size_query("trs_cin_glue_blocks", B)
        part = torch.empty(nblk, 2, E, dtype=torch.float32, device=g.device)
        call("trs_cin_glue_stats", ptr(g), B, P, E, value_dtype_code(g), ptr(part), stream_ptr())
        return part[:, 0].double().sum(0).to(g.dtype)
    gb = g.sum(0, dtype=torch.float32) if per_pair else g.reshape(-1, E).sum(0, dtype=torch.float32)
    return gb.to(g.dtype)


class _PairMul(Function):
    """out[b,p,:] = a[b,i_p,:] * c[b,j_p,:] + bias."""

    @staticmethod
    def forward(ctx, a, c, bias, bias_per_pair):
        require_device(a, c, bias)
        a, c = a.contiguous(), c.contiguous()
        B, N, E = a.shape
        bias_c = None if bias is None else bias.contiguous().to(a.dtype)
        out = torch.empty(B, N * (N - 1) // 2, E, dtype=a.dtype, device=a.device)
        call("trs_pair_mul_fwd", ptr(a), ptr(c), ptr(bias_c), int(bool(bias_per_pair)), B, N, E, value_dtype_code(a),
             ptr(out), stream_ptr())
        ctx.save_for_backward(a, c)
        ctx.bias_per_pair = bool(bias_per_pair)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, c = ctx.saved_tensors
        B, N, E = a.shape
        g = g.contiguous()
        ga, gc = torch.empty_like(a), torch.empty_like(c)
        call("trs_pair_mul_bwd", ptr(g), ptr(a), ptr(c), B, N, E, value_dtype_code(a), ptr(ga), ptr(gc), stream_ptr())
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _pair_bias_grad(g, ctx.bias_per_pair)
        return ga, gc, gb, None


def pair_mul(a: torch.Tensor, c: torch.Tensor, bias: Optional[torch.Tensor] = None, bias_per_pair: bool = False):
    if a.dim() != 3 or a.shape != c.shape:
        raise ValueError(f"pair_mul operands must both be (B, N, E), got {tuple(a.shape)} and {tuple(c.shape)}")
    return _PairMul.apply(a, c, bias, bias_per_pair)


class _RowsMulBias(Function):
    """out = a * c + bias on (B, P, E) operands that are already gathered per pair (trs_rows_mul_bias_fwd)."""

    @staticmethod
    def forward(ctx, a, c, bias, bias_per_pair):
        require_device(a, c, bias)
        B, P, E = a.shape
        a, c = a.contiguous(), c.contiguous()
        out = torch.empty_like(a)
        bias_c = None if bias is None else bias.to(a.dtype).contiguous()
        call("trs_rows_mul_bias_fwd", ptr(a), ptr(c), ptr(bias_c), int(bool(bias_per_pair)), B * P, P, E,
             value_dtype_code(a), ptr(out), stream_ptr())
        ctx.save_for_backward(a, c)
        ctx.bias_per_pair = bool(bias_per_pair)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, c = ctx.saved_tensors
        B, P, E = a.shape
        g = g.contiguous()
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        call("trs_rows_mul_bwd", ptr(g), ptr(a), ptr(c), B * P, E, value_dtype_code(a), ptr(ga), ptr(gc), stream_ptr())
        gb = _pair_bias_grad(g, ctx.bias_per_pair) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return ga, gc, gb, None


def rows_mul_bias(a: torch.Tensor, c: torch.Tensor, bias: Optional[torch.Tensor] = None, bias_per_pair: bool = False):
    """out = a * c + bias.  Operands (*, P, E) of equal shape -- any number of leading dimensions, as the reference's
    (N, *, H) contract allows (bilinear_interaction.py:11-79); they are folded into one batch axis for the kernel.
    bias: (E,) shared by every pair, or (P, E) with ``bias_per_pair``; its shape is checked here because the kernel
    indexes it without one."""
    if a.dim() < 2 or a.shape != c.shape:
        raise ValueError(f"rows_mul_bias operands must both be (*, P, E), got {tuple(a.shape)} and {tuple(c.shape)}")
    P, E = a.shape[-2], a.shape[-1]
    if bias is not None:
        want = (P, E) if bias_per_pair else (E,)
        if tuple(bias.shape) != want:
            raise ValueError(f"rows_mul_bias: bias must be {want} for operands {tuple(a.shape)}, got {tuple(bias.shape)}")
    if a.dim() == 3:
        return _RowsMulBias.apply(a, c, bias, bias_per_pair)
    lead = a.shape[:-2]
    out = _RowsMulBias.apply(a.reshape(-1, P, E), c.reshape(-1, P, E), bias, bias_per_pair)
    return out.reshape(*lead, P, E)


PAIR_GEMM_MIN_BATCH = 256      # below this the one-kernel path (weights streamed per sample group) is used


def _pair_gemm_route(x: torch.Tensor) -> bool:
    B, N, E = x.shape
    ve = 16 // x.element_size()
    vpr = E // ve if E % ve == 0 else 0
    return (B >= PAIR_GEMM_MIN_BATCH and vpr > 0 and (vpr & (vpr - 1)) == 0 and vpr <= 64
            and (N * E * 8 + N * N * 2 + 64) <= 64 * 1024)


def _pair_T(x: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """T[b,p,:] = x[b,i_p,:] @ W[p]: one plain GEMM per field i over its adjacent pairs (i, i+1..N-1)."""
    B, N, E = x.shape
    P = N * (N - 1) // 2
    T = torch.empty(B, P, E, dtype=x.dtype, device=x.device)
    T2 = T.view(B, P * E)
    for i in range(N - 1):
        p0, n_i = _pair_start(i, N), N - 1 - i
        Wcat = W[p0:p0 + n_i].permute(1, 0, 2).reshape(E, n_i * E)          # [e][(q,h)] = W[p0+q][e][h]
        torch.mm(x[:, i, :], Wcat, out=T2[:, p0 * E:(p0 + n_i) * E])
    return T


_pair_task_cache = {}


def _pair_tasks(N: int, device):
    """Task lists of the MFMA per-pair kernels, cached per (N, device):
    tasks_i (nti,3) = (i, j0, count <= 3): pairs (i, j0..j0+count-1);  seg_i (N+1): first task of field i;
    tasks_j (ntj,3) = (j, i0, count <= 3): pairs (i0..i0+count-1, j);  seg_j (N+1): first task of field j."""
    key = (N, str(device))
    t = _pair_task_cache.get(key)
    if t is None:
        ti, si, tj, sj = [], [0], [], [0]
        for i in range(N):
            ti += [(i, j0, min(3, N - j0)) for j0 in range(i + 1, N, 3)]
            si.append(len(ti))
        for j in range(N):
            tj += [(j, i0, min(3, j - i0)) for i0 in range(0, j, 3)]
            sj.append(len(tj))
        mk = lambda rows: torch.tensor(rows, dtype=torch.int32, device=device)
        t = _pair_task_cache[key] = (mk(ti), mk(si), mk(tj), mk(sj))
    return t


def _pair_mfma_fwd_ok(x: torch.Tensor) -> bool:
    return x.dtype == torch.bfloat16 and x.shape[2] in (32, 64) and x.shape[1] >= 2 and x.shape[0] >= 16


class _PairBilinear(Function):
    """T = x_i W_p;  mode 0: out[b,p] = sum_h T_h x_j[h]  |  mode 1: out[b,p,:] = T * x_j + bias_p.   W (P,E,E) [e][h].
    Two routes: one HIP kernel that streams W_p per group of samples (small batches, any shape), or -- at training
    batch sizes -- plain per-field GEMMs for T / dL/dx_i / dL/dW around HIP epilogue passes (trs_pair_epilogue_*)."""

    @staticmethod
    def forward(ctx, x, W, bias, mode):
        require_device(x, W, bias)
        x = x.contiguous()
        B, N, E = x.shape
        P = N * (N - 1) // 2
        if tuple(W.shape) != (P, E, E):
            raise ValueError(f"per-pair weights must be ({P}, {E}, {E}), got {tuple(W.shape)}")
        Wc = W.contiguous().to(x.dtype)
        bias_c = None if bias is None else bias.contiguous().to(x.dtype)
        ctx.gemm = _pair_gemm_route(x)
        if _pair_mfma_fwd_ok(x):
            # forward on the matrix cores in one kernel (W_p^T fragments resident per wave)
            Wt = Wc.transpose(1, 2).contiguous()
            tasks = _pair_tasks(N, x.device)[0]
            out = torch.empty((B, P) if mode == 0 else (B, P, E), dtype=x.dtype, device=x.device)
            call("trs_pair_bilinear_fwd_mfma", ptr(x), ptr(Wt), ptr(bias_c), ptr(tasks), tasks.shape[0], int(mode), B, N, E,
                 value_dtype_code(x), ptr(out), stream_ptr())
        elif ctx.gemm:
            T = _pair_T(x, Wc)
            out = torch.empty(B, P, dtype=x.dtype, device=x.device) if mode == 0 else None
            call("trs_pair_epilogue_fwd", ptr(T), ptr(x), ptr(bias_c), 1, int(mode), B, N, E, value_dtype_code(x),
                 ptr(out), stream_ptr())
            if mode == 1:
                out = T
        else:
            out = torch.empty((B, P) if mode == 0 else (B, P, E), dtype=x.dtype, device=x.device)
            call("trs_pair_bilinear_fwd", ptr(x), ptr(Wc), 1, ptr(bias_c), 1, int(mode), B, N, E, value_dtype_code(x),
                 ptr(out), stream_ptr())
        ctx.save_for_backward(x, Wc)
        ctx.mode = int(mode)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        B, N, E = x.shape
        P = N * (N - 1) // 2
        g = g.contiguous()
        need_w = ctx.needs_input_grad[1]
        if _pair_mfma_fwd_ok(x):
            # three MFMA kernels: dL/dx_i and dL/dx_j as per-task contribution rows + one reduction, dL/dW with K = batch
            ti, si, tj, sj = _pair_tasks(N, x.device)
            Wt = W.transpose(1, 2).contiguous()
            ci = torch.empty(B, ti.shape[0], E, dtype=x.dtype, device=x.device)
            cj = torch.empty(B, tj.shape[0], E, dtype=x.dtype, device=x.device)
            gx = torch.empty_like(x)
            call("trs_pair_bilinear_bwd_data_mfma", ptr(g), ptr(x), ptr(W), ptr(Wt), ptr(ti), ti.shape[0], ptr(si), ptr(tj),
                 tj.shape[0], ptr(sj), ctx.mode, B, N, E, value_dtype_code(x), ptr(ci), ptr(cj), ptr(gx), stream_ptr())
            gW = gb = None
            if need_w:
                gW = torch.empty(P, E, E, dtype=x.dtype, device=x.device)
                ws_bytes = size_query("trs_pair_bilinear_bwd_w_mfma_workspace_bytes", B, N, E)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
                call("trs_pair_bilinear_bwd_w_mfma", ptr(g), ptr(x), ctx.mode, B, N, E, value_dtype_code(x), ptr(gW),
                     ptr(ws), ws_bytes, stream_ptr())
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _pair_bias_grad(g, True)
            return gx, gW, gb, None
        if ctx.gemm:
            gT = _pair_T(x, W)                                  # recomputed, then overwritten by dL/dT in place
            gx = torch.empty_like(x)
            call("trs_pair_epilogue_bwd", ptr(g), ptr(x), ptr(gT), ctx.mode, B, N, E, value_dtype_code(x), ptr(gx),
                 stream_ptr())
            gT2 = gT.view(B, P * E)
            gxi = torch.empty(B, E, dtype=x.dtype, device=x.device)
            for i in range(N - 1):                              # dL/dx_i = dL/dT[:, pairs of i] @ Wcat_i^T
                p0, n_i = _pair_start(i, N), N - 1 - i
                Wcat = W[p0:p0 + n_i].permute(1, 0, 2).reshape(E, n_i * E)
                torch.mm(gT2[:, p0 * E:(p0 + n_i) * E], Wcat.t(), out=gxi)
                gx[:, i, :] += gxi
        else:
            gx = torch.empty_like(x)
            gT = torch.empty(B, P, E, dtype=x.dtype, device=x.device) if need_w else None
            call("trs_pair_bilinear_bwd_data", ptr(g), ptr(x), ptr(W), 1, ctx.mode, B, N, E, value_dtype_code(x), ptr(gx),
                 ptr(gT), stream_ptr())
        gW = gb = None
        if need_w:
            # dW[p] = sum_b x[b,i_p,:]^T gT[b,p,:]: for field i the pairs (i, i+1..N-1) are adjacent, so one plain GEMM
            # (E x B) @ (B x n_i*E) per field covers them (K = the batch; hipBLASLt)
            gW = torch.empty(P, E, E, dtype=x.dtype, device=x.device)
            gT2 = gT.view(B, P * E)
            for i in range(N - 1):
                p0, n_i = _pair_start(i, N), N - 1 - i
                blk = x[:, i, :].t() @ gT2[:, p0 * E:(p0 + n_i) * E]              # (E, n_i*E)
                gW[p0:p0 + n_i] = blk.view(E, n_i, E).permute(1, 0, 2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _pair_bias_grad(g, True)
        return gx, gW, gb, None


def pair_bilinear(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], mode: int) -> torch.Tensor:
    if x.dim() != 3:
        raise ValueError(f"pair-bilinear input must be (B, N, E), got {tuple(x.shape)}")
    return _PairBilinear.apply(x, W, bias, mode)


class _AFM(Function):
    """(out (B,E), attn (B,NC2)) = attention-weighted sum of the pair products; see trs_afm_fwd_dropout.  With a ``keep``
    mask (uint8 (B,NC2)) the reference's dropout on the scores (attentional_factorization_machine.py:82) happens inside
    the pass: ``attn`` is then the dropped scores (what the reference returns) and the sum uses them."""

    @staticmethod
    def forward(ctx, x, W1, b1, w2, b2, keep=None, keep_scale=1.0):
        require_device(x, W1, b1, w2, b2, keep)
        x = x.contiguous()
        B, N, E = x.shape
        A = W1.shape[0]
        P = N * (N - 1) // 2
        ps = [t.contiguous().to(x.dtype) for t in (W1, b1, w2.reshape(-1), b2.reshape(-1))]
        out = torch.empty(B, E, dtype=x.dtype, device=x.device)
        attn = torch.empty(B, P, dtype=x.dtype, device=x.device)
        attn_drop = None
        if keep is not None:
            if keep.dtype != torch.uint8 or tuple(keep.shape) != (B, P):
                raise ValueError(f"keep mask must be uint8 of shape {(B, P)}, got {keep.dtype} {tuple(keep.shape)}")
            keep = keep.contiguous()
            attn_drop = torch.empty_like(attn)
        call("trs_afm_fwd_dropout", ptr(x), ptr(ps[0]), ptr(ps[1]), ptr(ps[2]), ptr(ps[3]), ptr(keep), float(keep_scale),
             B, N, E, A, value_dtype_code(x), ptr(out), ptr(attn), ptr(attn_drop), stream_ptr())
        ctx.save_for_backward(x, attn, keep, *ps)
        ctx.keep_scale = float(keep_scale)
        ctx.set_materialize_grads(False)
        ctx.w2_shape, ctx.b2_shape = tuple(w2.shape), tuple(b2.shape)
        return out, (attn if keep is None else attn_drop)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_attn):
        x, attn, keep, W1, b1, w2, b2 = ctx.saved_tensors
        if g_out is None and g_attn is None:
            return None, None, None, None, None, None, None
        B, N, E = x.shape
        A = W1.shape[0]
        dev = x.device
        gx = torch.empty_like(x)
        gW1 = torch.zeros(A, E, dtype=torch.float32, device=dev)
        gv = torch.zeros(2 * A + 1, dtype=torch.float32, device=dev)
        ws_bytes = size_query("trs_afm_bwd_workspace_bytes", B, N, E, A)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        go = None if g_out is None else g_out.contiguous()
        ga = None if g_attn is None else g_attn.contiguous()
        call("trs_afm_bwd_dropout", ptr(go), ptr(ga), ptr(x), ptr(attn), ptr(keep), ctx.keep_scale, ptr(W1), ptr(b1),
             ptr(w2), B, N, E, A, value_dtype_code(x), ptr(gx), ptr(gW1), ptr(gv[:A]), ptr(gv[A:2 * A]), ptr(gv[2 * A:]),
             ptr(ws), ws_bytes, stream_ptr())
        dt = x.dtype
        return (gx, gW1.to(dt), gv[:A].to(dt), gv[A:2 * A].to(dt).reshape(ctx.w2_shape),
                gv[2 * A:].to(dt).reshape(ctx.b2_shape), None, None)


def afm(x: torch.Tensor, W1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
        keep: Optional[torch.Tensor] = None, keep_scale: float = 1.0):
    """``keep`` (uint8 (B,NC2), nonzero = kept) / ``keep_scale`` (1/(1-p)): dropout on the attention scores."""
    if x.dim() != 3:
        raise ValueError(f"AFM input must be (B, N, E), got {tuple(x.shape)}")
    return _AFM.apply(x, W1, b1, w2, b2, keep, keep_scale)


# --------------------------------------------------------------------------------------------
# K3: field-aware FM pair products
# --------------------------------------------------------------------------------------------
class _FFM(Function):
    @staticmethod
    def forward(ctx, x, num_fields):
        require_device(x)
        x = x.contiguous()
        B, NN, E = x.shape
        N = num_fields
        out = torch.empty(B, N * (N - 1) // 2, E, dtype=x.dtype, device=x.device)
        call("trs_ffm_fwd", ptr(x), B, N, E, value_dtype_code(x), ptr(out), stream_ptr())
        ctx.save_for_backward(x)
        ctx.N = N
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, NN, E = x.shape
        dx = torch.empty_like(x)
        call("trs_ffm_bwd", ptr(x), ptr(g.contiguous()), B, ctx.N, E, value_dtype_code(x), ptr(dx), stream_ptr())
        return dx, None


def ffm_layer(x: torch.Tensor, num_fields: int) -> torch.Tensor:
    if x.dim() != 3 or x.shape[1] != num_fields * num_fields:
        raise ValueError(f"FFM input must be (B, N*N={num_fields * num_fields}, E), got {tuple(x.shape)}")
    return _FFM.apply(x, num_fields)


# --------------------------------------------------------------------------------------------
# K4: cross network
# --------------------------------------------------------------------------------------------
class _Cross(Function):
    @staticmethod
    def forward(ctx, x, W, b, detach_first):
        require_device(x, W, b)
        x = x.contiguous()
        L, E, _ = W.shape
        rows = x.numel() // E
        Wc = W.to(x.dtype).contiguous()
        bc = b.to(x.dtype).contiguous()
        out = torch.empty_like(x)
        ws_bytes = size_query("trs_cross_workspace_bytes", rows, E, L, value_dtype_code(x))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
        call("trs_cross_fwd", ptr(x), ptr(Wc), ptr(bc), rows, E, L, value_dtype_code(x), ptr(out), ptr(ws), ws_bytes,
             stream_ptr())
        ctx.save_for_backward(x, Wc, bc)
        ctx.detach_first = bool(detach_first)
        ctx.param_dtypes = (W.dtype, b.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, Wc, bc = ctx.saved_tensors
        L, E, _ = Wc.shape
        rows = x.numel() // E
        dx = torch.empty_like(x)
        dW = torch.zeros(L, E, E, dtype=torch.float32, device=x.device)
        db = torch.zeros(L, E, dtype=torch.float32, device=x.device)
        ws_bytes = size_query("trs_cross_workspace_bytes", rows, E, L, value_dtype_code(x))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
        call("trs_cross_bwd", ptr(x), ptr(Wc), ptr(bc), ptr(g.contiguous()), rows, E, L, value_dtype_code(x),
             1 if ctx.detach_first else 0, ptr(dx), ptr(dW), ptr(db), ptr(ws), ws_bytes, stream_ptr())
        return dx, dW.to(ctx.param_dtypes[0]), db.to(ctx.param_dtypes[1]), None


def cross_network(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, detach_first: bool = True) -> torch.Tensor:
    """x: (..., E); W: (L,E,E) stacked nn.Linear weights; b: (L,E)."""
    if W.dim() != 3 or W.shape[1] != W.shape[2] or x.shape[-1] != W.shape[1] or b.shape != W.shape[:2]:
        raise ValueError(f"cross_network: bad shapes x{tuple(x.shape)} W{tuple(W.shape)} b{tuple(b.shape)}")
    return _Cross.apply(x, W, b, detach_first)


# --------------------------------------------------------------------------------------------
# K5: CIN contraction  y[b,c,e] = bias[c] + sum_{n,h} Wc[c,n*H+h] x0[b,n,e] xk[b,h,e]
# --------------------------------------------------------------------------------------------
class _CINContract(Function):
    @staticmethod
    def forward(ctx, x0, xk, Wc, bias):
        require_device(x0, xk, Wc, bias)
        x0 = x0.contiguous()
        xk = xk.contiguous()
        B, N, E = x0.shape
        H = xk.shape[1]
        C = Wc.shape[0]
        w = Wc.to(x0.dtype).contiguous()
        bb = None if bias is None else bias.to(x0.dtype).contiguous()
        y = torch.empty(B, C, E, dtype=x0.dtype, device=x0.device)
        call("trs_cin_fwd", ptr(x0), ptr(xk), ptr(w), ptr(bb), B, N, H, C, E, value_dtype_code(x0), ptr(y), ptr(None),
             stream_ptr())
        ctx.save_for_backward(x0, xk, w)
        ctx.meta = (Wc.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x0, xk, w = ctx.saved_tensors
        B, N, E = x0.shape
        H, C = xk.shape[1], w.shape[0]
        gy = gy.contiguous()
        need_x0, need_xk, need_w, need_b = ctx.needs_input_grad
        dW = torch.zeros(C, N * H, dtype=torch.float32, device=x0.device) if need_w else None
        dx0 = torch.empty_like(x0) if need_x0 else None
        dxk = torch.empty_like(xk) if need_xk else None
        call("trs_cin_bwd", ptr(x0), ptr(xk), ptr(w), ptr(gy), B, N, H, C, E, value_dtype_code(x0), ptr(dW), ptr(dx0),
             ptr(dxk), 0, stream_ptr())
        db = gy.float().sum(dim=(0, 2)).to(ctx.meta[1]) if (need_b and ctx.meta[1] is not None) else None
        return dx0, dxk, (dW.to(ctx.meta[0]) if need_w else None), db


def cin_contract(x0: torch.Tensor, xk: torch.Tensor, Wc: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x0 (B,N,E), xk (B,H,E), Wc (C,N*H) -> y (B,C,E); the (B,N*H,E) outer product is never materialised."""
    if x0.dim() != 3 or xk.dim() != 3 or x0.shape[0] != xk.shape[0] or x0.shape[2] != xk.shape[2]:
        raise ValueError(f"cin_contract: bad shapes x0{tuple(x0.shape)} xk{tuple(xk.shape)}")
    if Wc.dim() != 2 or Wc.shape[1] != x0.shape[1] * xk.shape[1]:
        raise ValueError(f"cin_contract: Wc must be (C, N*H={x0.shape[1] * xk.shape[1]}), got {tuple(Wc.shape)}")
    return _CINContract.apply(x0, xk, Wc, bias)


# --------------------------------------------------------------------------------------------
# K5, channels-last (MFMA) form: x0T (B,E,ld0), xkT (B,E,>=H) view, -> yT (B,E,C)
# --------------------------------------------------------------------------------------------
def cin_cl_supported(x: torch.Tensor, out_channels: Sequence[int], hidden_sizes: Sequence[int]) -> bool:
    """Whole-stack check for the channels-last MFMA path: bf16 on device, E % 16 == 0, every layer's
    C % 32 == 0, field count N <= 256, hidden widths multiples of 32 in {32,64,128,256}."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 16 == 0 and x.shape[1] <= 256):
        return False
    if any(c % 32 != 0 for c in out_channels):
        return False
    return all(h in (32, 64, 128, 256) for h in hidden_sizes)


CIN_FOLD_SYMMETRIC = os.environ.get("TRS_CIN_FOLD", "1") != "0"
# the last CIN layer's backward leaves out the output channels that provably carry no gradient (see _CINContractCL.forward)
CIN_SKIP_DEAD = os.environ.get("TRS_CIN_SKIP_DEAD", "1") != "0"


def cin_fold_symmetric(Wc: torch.Tensor, N: int) -> torch.Tensor:
    """First CIN layer (xk IS x0, compress_interaction_network.py:125-132 with H_0 = N): the products x0[n]*x0[h] are
    symmetric in (n, h), so the weight of the pair lives on h <= n as W[n,h] + W[h,n] (fp32 sum) and is 0 above the
    diagonal.  Wc (C, N*N) -> (C, N*N) fp32; sum_{n,h} W x0[n] x0[h] is unchanged."""
    C = Wc.shape[0]
    W3 = Wc.float().view(C, N, N)
    return (torch.tril(W3) + torch.triu(W3, 1).transpose(1, 2)).reshape(C, N * N)


class _CINContractCL(Function):
    @staticmethod
    def forward(ctx, x0T, xkT, Wc, bias, N, H, x0_cf=None, xk_cf=None, live=None):
        require_device(x0T, xkT, Wc, bias)
        B, E, ld0 = x0T.shape
        ldk = xkT.stride(1)
        C = Wc.shape[0]
        if xkT.stride(2) != 1 or xkT.stride(0) != E * ldk or x0T.stride() != (E * ld0, ld0, 1):
            raise ValueError("cin_contract_cl: x0T must be contiguous and xkT a (B,E,>=H) view with unit inner stride")
        # First layer (xk IS x0): the products x0[n]*x0[h] are symmetric in (n,h), so W[n,h] + W[h,n] folded onto h <= n
        # (one fp32 add, one bf16 rounding) gives the same sums from half the weights; the kernels skip the k-steps /
        # h tiles that then hold only zeros.  Pays once a field needs more than one 32-wide k-step.
        tri = int(xkT.data_ptr() == x0T.data_ptr() and ldk == ld0 and H == N and N > 32 and CIN_FOLD_SYMMETRIC)
        if tri:
            w = cin_fold_symmetric(Wc.detach(), N).to(torch.bfloat16)
        else:
            w = Wc.to(torch.bfloat16).contiguous()
        bb = None if bias is None else bias.to(torch.bfloat16).contiguous()
        yT = torch.empty(B, E, C, dtype=torch.bfloat16, device=x0T.device)
        ws_bytes = size_query("trs_cin_cl_workspace_bytes", N, H, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x0T.device)
        call("trs_cin_cl_fwd", ptr(x0T), ld0, ptr(xkT), ldk, ptr(w), ptr(bb), B, N, H, C, E, _abi.TRS_BF16, tri, ptr(yT),
             ptr(ws), ws_bytes, stream_ptr())
        ctx.tri = tri
        # live: the caller's promise that the gradient of the output channels [live, C) is exactly zero (the LAST layer's
        # "hidden" half, which the reference computes, splits off and never uses -- compress_interaction_network.py:151-156,
        # 176-181): the backward then leaves their (zero) terms out of both contractions
        ctx.live = int(live) if (live is not None and 0 < int(live) < C and not tri and CIN_SKIP_DEAD) else None
        ctx.save_for_backward(x0T, xkT, w)
        ctx.x0_cf = x0_cf        # the caller's channels-first (B,N,E) input, if it has one (saves a transpose per layer)
        ctx.xk_cf = xk_cf        # likewise the hidden state (B,H,E), emitted by the glue pass of the previous layer
        ctx.meta = (N, H, Wc.dtype, None if bias is None else bias.dtype)
        return yT

    @staticmethod
    @once_differentiable
    def backward(ctx, gyT):
        x0T, xkT, w = ctx.saved_tensors
        N, H, wdt, bdt = ctx.meta
        B, E, _ = x0T.shape
        C = w.shape[0]
        need_x0, need_xk, need_w, need_b = ctx.needs_input_grad[:4]
        gy_cf = getattr(gyT, '_trs_cf', None)      # (B,C,E) copy emitted by the glue backward, if any
        gyT = gyT.contiguous()
        dev = x0T.device
        ld0, ldk = x0T.shape[2], xkT.stride(1)
        db = None
        pre = getattr(gyT, '_trs_colsum', None)    # column sums emitted by the glue backward that produced gyT
        if need_b and bdt is not None:
            if pre is not None and pre[1] == gyT._version:
                db = pre[0].double().sum(0).to(bdt)
            elif cin_glue_supported(gyT, 0, 0):
                # column sums of the (B,E,C) gradient by the glue statistics kernel (one bf16 read, fp32 partials)
                nblk = size_query("trs_cin_glue_blocks", B)
                part = torch.empty(nblk, 2, C, dtype=torch.float32, device=gyT.device)
                call("trs_cin_glue_stats", ptr(gyT), B, E, C, value_dtype_code(gyT), ptr(part), stream_ptr())
                db = part[:, 0].double().sum(0).to(bdt)
            else:
                db = gyT.float().sum(dim=(0, 1)).to(bdt)
        dx0T = dxkT = dW = None
        mfma_data = C in (32, 64, 128, 256)
        mfma_dw = C in (64, 128, 256) and E in (32, 64, 128)
        Cl = ctx.live
        if Cl is not None and not (Cl in (64, 128, 256) and mfma_data and mfma_dw and (gy_cf is None or gy_cf.is_contiguous())):
            Cl = None
        if Cl is not None:
            # the channels [Cl, C) carry no gradient: contraction over the first Cl channels only, the rest of dW stays zero
            if need_x0 or need_xk:
                ldo = ((H + 31) // 32) * 32
                dx0T = torch.empty_like(x0T)
                dxk_pad = torch.empty(B, E, ldo, dtype=torch.bfloat16, device=dev)
                ws_bytes = size_query("trs_cin_cl_bwd_data_workspace_bytes", N, H, Cl)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                call("trs_cin_cl_bwd_data_live", ptr(x0T), ld0, ptr(xkT), ldk, ptr(gyT), C, ptr(w), B, N, H, Cl, E,
                     _abi.TRS_BF16, ptr(dx0T), ptr(dxk_pad), ldo, ptr(ws), ws_bytes, stream_ptr())
                if tuple(dxk_pad.shape) == tuple(xkT.shape):
                    dxkT = dxk_pad
                else:
                    dxkT = torch.zeros(xkT.shape, dtype=xkT.dtype, device=dev)
                    dxkT[:, :, :H] = dxk_pad[:, :, :H]
            if need_w:
                x0_cf, xk_cf = ctx.x0_cf, ctx.xk_cf
                x0 = x0_cf if (x0_cf is not None and tuple(x0_cf.shape) == (B, N, E) and x0_cf.is_contiguous()) \
                    else x0T[:, :, :N].transpose(1, 2).contiguous()
                xk = xk_cf if (xk_cf is not None and tuple(xk_cf.shape) == (B, H, E) and xk_cf.is_contiguous()) \
                    else xkT[:, :, :H].transpose(1, 2).contiguous()
                gy = gy_cf if (gy_cf is not None and tuple(gy_cf.shape) == (B, C, E)) else gyT.transpose(1, 2).contiguous()
                dW = torch.zeros(C, N * H, dtype=torch.float32, device=dev)
                ws_bytes = size_query("trs_cin_dw_workspace_bytes", B, N, H, Cl)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                call("trs_cin_dw_live", ptr(gy), C * E, ptr(x0), ptr(xk), B, N, H, Cl, E, _abi.TRS_BF16, ptr(dW), ptr(ws),
                     ws_bytes, stream_ptr())
            return (dx0T if need_x0 else None, dxkT if need_xk else None, (dW.to(wdt) if need_w else None), db, None, None,
                    None, None, None)
        if need_x0 or need_xk:
            if mfma_data:
                ldo = ((H + 31) // 32) * 32
                dx0T = torch.empty_like(x0T)
                # tri: xkT is x0T, the kernel returns the sum of the two gradients (= the gradient of the quadratic form
                # under the folded weights = under the original ones) in dx0T and xkT's slot of this function stays None
                dxk_pad = None if ctx.tri else torch.empty(B, E, ldo, dtype=torch.bfloat16, device=dev)
                ws_bytes = size_query("trs_cin_cl_bwd_data_workspace_bytes", N, H, C)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                call("trs_cin_cl_bwd_data", ptr(x0T), ld0, ptr(xkT), ldk, ptr(gyT), ptr(w), B, N, H, C, E, _abi.TRS_BF16,
                     ctx.tri, ptr(dx0T), ptr(dxk_pad), ldo, ptr(ws), ws_bytes, stream_ptr())
                if ctx.tri:
                    dxkT = None
                elif tuple(dxk_pad.shape) == tuple(xkT.shape):     # already zero past H (zero weight fragments)
                    dxkT = dxk_pad
                else:
                    dxkT = torch.zeros(xkT.shape, dtype=xkT.dtype, device=dev)
                    dxkT[:, :, :H] = dxk_pad[:, :, :H]
        if (need_x0 or need_xk) and not mfma_data or (need_w and not mfma_dw):
            # generic channels-first kernels for shapes the MFMA kernels do not cover
            x0 = x0T[:, :, :N].transpose(1, 2).contiguous()
            xk = xkT[:, :, :H].transpose(1, 2).contiguous()
            gy = gyT.transpose(1, 2).contiguous()
            gW = torch.zeros(C, N * H, dtype=torch.float32, device=dev) if (need_w and not mfma_dw) else None
            gx0 = torch.empty_like(x0) if not mfma_data else None
            gxk = torch.empty_like(xk) if not mfma_data else None
            call("trs_cin_bwd", ptr(x0), ptr(xk), ptr(w), ptr(gy), B, N, H, C, E, _abi.TRS_BF16, ptr(gW), ptr(gx0),
                 ptr(gxk), 0, stream_ptr())
            if not mfma_data:
                dx0T = torch.zeros_like(x0T)
                dx0T[:, :, :N] = gx0.transpose(1, 2)
                dxkT = torch.zeros(xkT.shape, dtype=xkT.dtype, device=dev)
                dxkT[:, :, :H] = gxk.transpose(1, 2)
            if gW is not None:
                dW = gW
        if need_w and mfma_dw:
            x0_cf = ctx.x0_cf
            if x0_cf is not None and tuple(x0_cf.shape) == (B, N, E) and x0_cf.is_contiguous():
                x0 = x0_cf
            else:
                x0 = x0T[:, :, :N].transpose(1, 2).contiguous()      # channels-first copies: pixels contiguous
            xk_cf = ctx.xk_cf
            if xkT.data_ptr() == x0T.data_ptr() and H == N:
                xk = x0                                               # first layer: xk is x0
            elif xk_cf is not None and tuple(xk_cf.shape) == (B, H, E) and xk_cf.is_contiguous():
                xk = xk_cf
            else:
                xk = xkT[:, :, :H].transpose(1, 2).contiguous()
            if gy_cf is not None and tuple(gy_cf.shape) == (B, C, E) and gy_cf.is_contiguous():
                gy = gy_cf
            else:
                gy = gyT.transpose(1, 2).contiguous()
            dW = torch.zeros(C, N * H, dtype=torch.float32, device=dev)
            ws_bytes = size_query("trs_cin_dw_workspace_bytes", B, N, H, C)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            call("trs_cin_dw", ptr(gy), ptr(x0), ptr(xk), B, N, H, C, E, _abi.TRS_BF16, int(ctx.tri and xk is x0), ptr(dW),
                 ptr(ws), ws_bytes, stream_ptr())
        return (dx0T if need_x0 else None, dxkT if (need_xk and dxkT is not None) else None,
                (dW.to(wdt) if need_w else None), db, None, None,
                None, None, None)


def cin_contract_cl(x0T: torch.Tensor, xkT: torch.Tensor, Wc: torch.Tensor, bias: Optional[torch.Tensor], N: int,
                    H: int, x0_cf: Optional[torch.Tensor] = None, xk_cf: Optional[torch.Tensor] = None,
                    live: Optional[int] = None) -> torch.Tensor:
    """channels-last CIN contraction on the matrix cores: x0T (B,E,ld0>=N, zero padded), xkT (B,E,>=H view).
    ``x0_cf`` / ``xk_cf``: channels-first copies (B,N,E) / (B,H,E) of the same values when the caller has them.
    ``live``: the caller's promise that only the first ``live`` output channels can receive a gradient (the rest feed
    nothing downstream): the backward leaves the others' exactly-zero terms out."""
    return _CINContractCL.apply(x0T, xkT, Wc, bias, N, H, None if x0_cf is None else x0_cf.detach(),
                                None if xk_cf is None else xk_cf.detach(), live)


# --------------------------------------------------------------------------------------------
# K3 fused: field-aware lookup + pair products straight from the N tables
# --------------------------------------------------------------------------------------------
class _FFMFused(Function):
    @staticmethod
    def forward(ctx, idx, offsets, *weights):
        require_device(idx, offsets, *weights)
        B, N = idx.shape
        if len(weights) != N:
            raise ValueError(f"need {N} tables, got {len(weights)}")
        V, E = weights[0].shape
        ws = [w.contiguous() for w in weights]
        out = torch.empty(B, N * (N - 1) // 2, E, dtype=ws[0].dtype, device=ws[0].device)
        flag = _ErrFlag(out.device)
        call("trs_ffm_fused_fwd", ptr(_pointer_table(ws)), V, E, value_dtype_code(ws[0]), ptr(idx), index_dtype_code(idx),
             ptr(offsets), B, N, ptr(out), ptr(flag.t), stream_ptr())
        flag.check("ffm_fused_fwd")
        if any(ctx.needs_input_grad[2:]):
            prefetch_row_buckets(idx, offsets, V)
        ctx.save_for_backward(idx, offsets, *weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        idx, offsets, *weights = ctx.saved_tensors
        B, N = idx.shape
        V, E = weights[0].shape
        ws = [w.contiguous() for w in weights]
        rb = row_buckets(idx, offsets, V)
        grads = [torch.empty_like(w) for w in ws]
        # the global row ids transposed to (N, B) int32: table i's walk reads ONE 4 B-per-sample column (cache-resident)
        # instead of a random 8-byte load per lookup (9.5 -> see profiles/r05_kernels.md)
        rid_t = None
        if V < 2 ** 31 and B > 0:
            # range check BEFORE narrowing: an id such as 2^32 + 5 must stay out of range (-1: contributes nothing, as in
            # the forward), not wrap onto a foreign row
            rid = idx.long() if offsets is None else idx.long() + offsets
            rid = torch.where((rid >= 0) & (rid < V), rid, rid.new_full((), -1))
            rid_t = rid.t().contiguous().to(torch.int32)
        call("trs_ffm_fused_bwd", ptr(_pointer_table(ws)), V, E, value_dtype_code(ws[0]), ptr(idx), index_dtype_code(idx),
             ptr(offsets), ptr(g.contiguous()), ptr(rb.row_start), ptr(rb.perm), B, N, ptr(_pointer_table(grads)),
             ptr(rid_t), stream_ptr())
        return (None, None, *grads)


def ffm_fused(weights: Sequence[torch.Tensor], idx: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """(B,N) indices + N field-aware tables -> (B, N(N-1)/2, E) pair products; (B,N*N,E) is never materialised."""
    idx = _as_index(idx)
    return _FFMFused.apply(idx, offsets, *weights)


# --------------------------------------------------------------------------------------------
# MLP backward epilogue: relu backward + bias gradient in one pass (GEMMs stay on hipBLASLt)
# --------------------------------------------------------------------------------------------
def rowdot_supported(h2: torch.Tensor) -> bool:
    """one-output Linear as a row-wise dot product: rows of 16-byte vectors, a power of two of them, at most 64"""
    if not (h2.is_cuda and h2.dim() == 2 and h2.dtype in (torch.float32, torch.bfloat16) and h2.is_contiguous()):
        return False
    ve = 16 // h2.element_size()
    lpr = h2.shape[1] // ve if h2.shape[1] % ve == 0 else 0
    return 1 <= lpr <= 64 and (lpr & (lpr - 1)) == 0


def rowdot_width_supported(C: int, elem_size: int) -> bool:
    lpr = C * elem_size // 16 if (C * elem_size) % 16 == 0 else 0
    return 1 <= lpr <= 64 and (lpr & (lpr - 1)) == 0


class _RowDot(Function):
    """out (rows, 1) = h (rows, C) @ w (1, C)^T + b (1): trs_rowdot_fwd / trs_rowdot_bwd.  ``w_use`` / ``b_use`` are
    the (possibly zero-padded) tensors the kernels read; gradients are returned for ``weight`` / ``bias``."""

    @staticmethod
    def forward(ctx, h, weight, bias, w_use, b_use):
        W = (weight if w_use is None else w_use).reshape(-1).contiguous()
        Bv = bias if w_use is None else b_use
        h2 = h.reshape(-1, h.shape[-1])
        out = torch.empty(h2.shape[0], 1, dtype=h.dtype, device=h.device)
        call("trs_rowdot_fwd", ptr(h2), ptr(W), ptr(Bv), h2.shape[0], h2.shape[1], value_dtype_code(h2), ptr(out),
             stream_ptr())
        ctx.save_for_backward(h2, W)
        ctx.meta = (tuple(h.shape), tuple(weight.shape), weight.dtype, bias is not None)
        return out.reshape(*h.shape[:-1], 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        h2, W = ctx.saved_tensors
        hshape, wshape, wdt, has_bias = ctx.meta
        rows, C = h2.shape
        g2 = g.reshape(-1).contiguous()
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        gh = torch.empty_like(h2) if need_h else None
        gw = torch.empty(C + 1, dtype=torch.float32, device=h2.device) if need_w else None
        ws_bytes = size_query("trs_rowdot_bwd_workspace_bytes", rows, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=h2.device) if need_w else None
        call("trs_rowdot_bwd", ptr(g2), ptr(h2), ptr(W), rows, C, value_dtype_code(h2), ptr(gh),
             ptr(gw), ptr(gw[C:]) if need_w else ptr(None), ptr(ws), ws_bytes if need_w else 0, stream_ptr())
        gq = gw.to(wdt) if need_w else None                      # one cast for weight row and bias
        g_weight = gq[:wshape[1]].reshape(wshape) if (need_w and ctx.needs_input_grad[1]) else None
        g_bias = gq[C:] if (need_w and has_bias and ctx.needs_input_grad[2]) else None
        return (gh.reshape(hshape) if need_h else None), g_weight, g_bias, None, None


def transpose_pad_supported(x: torch.Tensor, ld: int) -> bool:
    """(B, N, E) bf16 HIP tensor -> (B, E, ld) with zeros behind the N fields in one pass (trs_transpose_pad)"""
    return (x.is_cuda and x.dim() == 3 and x.dtype == torch.bfloat16 and x.shape[1] <= ld <= 64 and ld % 8 == 0
            and x.shape[2] <= 64 and x.shape[2] % 8 == 0)


class _TransposePad(Function):
    """x (B,N,E) -> (B,E,ld): out[b,e,n] = x[b,n,e], zeros for n >= N; the gradient is the transposition back.  What
    ``x.new_zeros(B,E,ld)[:, :, :N] = x.transpose(1,2)`` does in a fill + a strided copy forward and a clone + a slice
    clone backward (compress_interaction_network.py:105 keeps the activations as ('B','E','N'))."""

    @staticmethod
    def forward(ctx, x, ld):
        x = x.contiguous()
        B, N, E = x.shape
        out = torch.empty(B, E, ld, dtype=x.dtype, device=x.device)
        call("trs_transpose_pad", ptr(x), B, N, E, E, ptr(out), ld, value_dtype_code(x), stream_ptr())
        ctx.dims = (N, E, ld)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        N, E, ld = ctx.dims
        g = g.contiguous()
        gx = torch.empty(g.shape[0], N, E, dtype=g.dtype, device=g.device)
        call("trs_transpose_pad", ptr(g), g.shape[0], E, N, ld, ptr(gx), E, value_dtype_code(g), stream_ptr())
        return gx, None


def transpose_pad(x: torch.Tensor, ld: int) -> torch.Tensor:
    return _TransposePad.apply(x, ld)


def cat_head_supported(a: torch.Tensor, d: torch.Tensor, weight: torch.Tensor) -> bool:
    """one-output Linear over cat((a, d), dim=2).flatten(1) as one pass over the two blocks (trs_cat_head_fwd / _bwd):
    (B, N, Ea) and (B, N, Eb) HIP tensors of one dtype (bf16 / fp32) with rows of whole 16-byte vectors, at most 1024
    vectors per sample, weight (1, N * (Ea + Eb))"""
    if not (a.is_cuda and d.is_cuda and a.dim() == 3 and d.dim() == 3 and a.shape[:2] == d.shape[:2]
            and a.dtype == d.dtype == weight.dtype and a.dtype in (torch.float32, torch.bfloat16)):
        return False
    N, Ea, Eb = a.shape[1], a.shape[2], d.shape[2]
    ve = 16 // a.element_size()
    return (Ea % ve == 0 and Eb % ve == 0 and N * (Ea + Eb) // ve <= 1024
            and tuple(weight.shape) == (1, N * (Ea + Eb)))


class _CatHead(Function):
    """out (B, 1) = cat((a, d), dim=2).flatten(1) @ weight (1, N*(Ea+Eb))^T + bias: the head of the reference's
    DeepAndCrossNetworkModel (deep_and_cross_network.py:82-92) without the (B, N, Ea+Eb) concatenation, its slice
    copies in the backward and the 4992-wide GEMV / GEMMs around them."""

    @staticmethod
    def forward(ctx, a, d, weight, bias):
        a, d = a.contiguous(), d.contiguous()
        W = weight.reshape(-1).contiguous()
        B, N, Ea = a.shape
        Eb = d.shape[2]
        out = torch.empty(B, 1, dtype=a.dtype, device=a.device)
        call("trs_cat_head_fwd", ptr(a), ptr(d), ptr(W), ptr(bias), B, N, Ea, Eb, value_dtype_code(a), ptr(out),
             stream_ptr())
        ctx.save_for_backward(a, d, W)
        ctx.meta = (tuple(weight.shape), bias is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, d, W = ctx.saved_tensors
        wshape, has_bias = ctx.meta
        B, N, Ea = a.shape
        Eb = d.shape[2]
        C = N * (Ea + Eb)
        g2 = g.reshape(-1).contiguous()
        need_a, need_d = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_w = ctx.needs_input_grad[2] or (has_bias and ctx.needs_input_grad[3])
        ga = torch.empty_like(a) if need_a else None
        gd = torch.empty_like(d) if need_d else None
        gw = torch.empty(C + 1, dtype=torch.float32, device=a.device) if need_w else None
        ws_bytes = size_query("trs_cat_head_bwd_workspace_bytes", B, N, Ea, Eb) if need_w else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if need_w else None
        call("trs_cat_head_bwd", ptr(g2), ptr(a), ptr(d), ptr(W), B, N, Ea, Eb, value_dtype_code(a), ptr(ga), ptr(gd),
             ptr(gw), ptr(gw[C:]) if need_w else ptr(None), ptr(ws), ws_bytes, stream_ptr())
        gq = gw.to(W.dtype) if need_w else None                  # one cast for the weight row and the bias
        g_weight = gq[:C].reshape(wshape) if (need_w and ctx.needs_input_grad[2]) else None
        g_bias = gq[C:] if (need_w and has_bias and ctx.needs_input_grad[3]) else None
        return ga, gd, g_weight, g_bias


def cat_head(a: torch.Tensor, d: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``F.linear(torch.cat((a, d), dim=2).flatten(1), weight, bias)`` for a one-output Linear; see cat_head_supported"""
    return _CatHead.apply(a, d, weight, bias)


def relu_bwd_bias_supported(y: torch.Tensor) -> bool:
    row_bytes = y.shape[-1] * y.element_size()
    return (y.is_cuda and y.dtype in (torch.float32, torch.bfloat16) and y.is_contiguous()
            and row_bytes % 16 == 0 and row_bytes <= 4096)


def relu_bwd_bias(gy: torch.Tensor, y: torch.Tensor):
    """gz = gy * (y > 0), gb = gz.sum(0) (fp32) for y = relu(...) of shape (rows, C)."""
    rows, C = y.shape
    gy = gy.contiguous()
    gz = torch.empty_like(gy)
    gb = torch.empty(C, dtype=torch.float32, device=y.device)
    ws_bytes = size_query("trs_relu_bwd_bias_workspace_bytes", rows, C)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=y.device)
    call("trs_relu_bwd_bias", ptr(gy), ptr(y), rows, C, value_dtype_code(y), ptr(gz), ptr(gb), ptr(ws), ws_bytes,
         stream_ptr())
    return gz, gb


# --------------------------------------------------------------------------------------------
# N4: per-field MLP (Linear -> ReLU ... -> Linear on every row of a (B,N,E) block) as one kernel per direction
# --------------------------------------------------------------------------------------------
FUSED_MLP = os.environ.get("TRS_FUSED_MLP", "1") not in ("", "0")
FUSED_MLP_MIN_ROWS = 4096


def _pad32(v: int) -> int:
    return (v + 31) // 32 * 32


def _i32_array(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


_mlp_fused_verdicts = {}


def mlp_fused_supported_for(rows: int, dtype: torch.dtype, is_cuda: bool, widths: Sequence[int]) -> bool:
    """bf16 rows on the HIP device, every width a multiple of 8 and at most 512, enough rows to fill the chip.  Takes the
    row count / dtype / placement instead of a tensor (no throwaway allocation) and remembers the library's verdict per
    width list."""
    if not (FUSED_MLP and is_cuda and dtype == torch.bfloat16 and rows >= FUSED_MLP_MIN_ROWS):
        return False
    if len(widths) < 2 or len(widths) > 9 or any(w % 8 or w > 512 for w in widths):
        return False
    key = tuple(int(w) for w in widths)
    ok = _mlp_fused_verdicts.get(key)
    if ok is None:
        ok = _mlp_fused_verdicts[key] = bool(_abi.load().trs_mlp_fused_supported(len(widths) - 1, _i32_array(widths)))
    return ok


def mlp_fused_supported(x: torch.Tensor, widths: Sequence[int]) -> bool:
    return mlp_fused_supported_for(x.numel() // max(1, x.shape[-1]), x.dtype, x.is_cuda, widths)


MLP_FAMILY_AUTO, MLP_FAMILY_TILE, MLP_FAMILY_ROW_OWNER, MLP_FAMILY_MIXED = 0, 1, 2, 3
_mlp_family_request = MLP_FAMILY_AUTO


class mlp_family:
    """``with mlp_family(MLP_FAMILY_TILE | MLP_FAMILY_ROW_OWNER | MLP_FAMILY_AUTO):`` -- the kernel family the fused-MLP
    FORWARDS issued inside the block ask for (a request the library resolves per call, trs_mlp_fused_family; a family
    that does not cover a stack falls back to AUTO for it).  Backwards never look at this: every autograd node records the
    family its forward ran and hands that to trs_mlp_fused_bwd_data."""

    def __init__(self, request: int):
        self.request = int(request)

    def __enter__(self):
        global _mlp_family_request
        self.prev, _mlp_family_request = _mlp_family_request, self.request
        return self

    def __exit__(self, *exc):
        global _mlp_family_request
        _mlp_family_request = self.prev
        return False


def mlp_fused_family(widths: Sequence[int], rows: int, request: Optional[int] = None) -> int:
    """The kernel family (MLP_FAMILY_TILE / MLP_FAMILY_ROW_OWNER) a forward of this stack runs"""
    req = _mlp_family_request if request is None else int(request)
    wl = _i32_array(widths)
    fam = int(_abi.load().trs_mlp_fused_family(len(widths) - 1, wl, int(rows), req))
    if fam == 0 and req != MLP_FAMILY_AUTO:
        fam = int(_abi.load().trs_mlp_fused_family(len(widths) - 1, wl, int(rows), MLP_FAMILY_AUTO))
    if fam == 0:
        raise RuntimeError(f"torecsys_amd: no fused-MLP kernel family for widths {list(widths)}")
    return fam


MLP_PHASE_ALL, MLP_PHASE_PACK, MLP_PHASE_RUN = 0, 1, 2


def fused_mlp_pack(Ws: Sequence[torch.Tensor], bs: Optional[Sequence[torch.Tensor]], widths: Sequence[int], rows: int,
                   family: int, backward: bool) -> torch.Tensor:
    """The PACK phase of trs_mlp_fused_fwd / _bwd_data on the current stream: the weights in MFMA fragment order (and the
    zeroed partial sums) in a fresh workspace, which the matching ``fused_mlp_*_raw(packed_ws=...)`` call then runs on.
    Depends on the parameters only -- callers enqueue it on a side stream while the layer in front of the stack runs."""
    L = len(Ws)
    wl = _i32_array(widths)
    ws_bytes = size_query("trs_mlp_fused_workspace_bytes", L, wl)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Ws[0].device)
    if backward:
        call("trs_mlp_fused_bwd_data", ptr(None), rows, L, wl, _ptr_array(Ws), ptr(None), ptr(None), ptr(None), ptr(None),
             ptr(None), ptr(None), _abi.TRS_BF16, int(family), MLP_PHASE_PACK, ptr(ws), ws_bytes, stream_ptr())
    else:
        call("trs_mlp_fused_fwd", ptr(None), rows, L, wl, _ptr_array(Ws), _ptr_array(bs), ptr(None), ptr(None), ptr(None),
             ptr(None), _abi.TRS_BF16, int(family), MLP_PHASE_PACK, 0, ptr(ws), ws_bytes, stream_ptr())
    return ws


def fused_mlp_pack_branch(Ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor], widths: Sequence[int], rows: int,
                          family: int, backward: bool, gemm: Optional[tuple] = None):
    """(ws_fwd, ws_bwd | None, ws_gemm | None): what fused_mlp_pack(forward), fused_mlp_pack(backward) and
    rows_gemm_pack(*gemm) -- gemm = (W, x_stride, out_f, in_f) -- leave, on the current stream.  The tile family does it
    in ONE launch (trs_mlp_pack_branch), the row-owner family in one per workspace."""
    if family != MLP_FAMILY_TILE:
        return (fused_mlp_pack(Ws, bs, widths, rows, family, False),
                fused_mlp_pack(Ws, None, widths, rows, family, True) if backward else None,
                rows_gemm_pack(gemm[0], rows, gemm[1], gemm[2], gemm[3]) if gemm is not None else None)
    L, dev = len(Ws), Ws[0].device
    wl = _i32_array(widths)
    ws_bytes = size_query("trs_mlp_fused_workspace_bytes", L, wl)
    ws_f = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    ws_b = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if backward else None
    ws_g, g_bytes, gW, g_out, g_in = None, 0, None, 0, 0
    if gemm is not None:
        gW, _, g_out, g_in = gemm
        g_bytes = size_query("trs_rows_gemm_workspace_bytes", g_out, g_in)
        ws_g = torch.empty(g_bytes, dtype=torch.uint8, device=dev)
    call("trs_mlp_pack_branch", rows, L, wl, _ptr_array(Ws), _ptr_array(bs), int(family), ptr(ws_f), ptr(ws_b), ws_bytes,
         ptr(gW), g_out, g_in, ptr(ws_g), g_bytes, stream_ptr())
    return ws_f, ws_b, ws_g


def fused_mlp_forward_raw(x2: torch.Tensor, Ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor],
                          input_mask: bool = False, family: Optional[int] = None,
                          packed_ws: Optional[torch.Tensor] = None, x_stride: int = 0):
    """trs_mlp_fused_fwd on rows x2 (rows, widths[0]): returns (y (rows, widths[L]), hidden [(rows, pad32(w))] -- the
    ReLU outputs of the hidden layers, zero in the padding columns --, masks [the sign bits of the hidden layers in the
    kernel's own order: opaque bytes for trs_mlp_fused_bwd_data], [with ``input_mask`` (x2 is itself a ReLU output): the
    sign bits of x2 in the same form,] family -- the kernel family that ran, which the backward of THESE masks must be
    given (the two families lay the sign bits out differently)."""
    L = len(Ws)
    widths = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
    rows, dev = x2.shape[0], x2.device
    # ``x_stride`` (row-owner / mixed family): x2's rows are wider than the stack's input -- the first widths[0] columns
    # of every row are read
    if x_stride and (x_stride < widths[0] or x2.shape[1] != x_stride or not x2.is_contiguous()):
        raise ValueError(f"fused_mlp_forward_raw: x_stride {x_stride} does not describe x2 {tuple(x2.shape)}")
    fam = mlp_fused_family(widths, rows, family)
    hidden = [torch.empty(rows, _pad32(widths[l + 1]), dtype=torch.bfloat16, device=dev) for l in range(L - 1)]
    mask_bytes = size_query("trs_mlp_fused_mask_bytes", rows)
    masks = [torch.empty(mask_bytes, dtype=torch.uint8, device=dev) for _ in range(L - 1)]
    mask_in = torch.empty(mask_bytes, dtype=torch.uint8, device=dev) if input_mask else None
    y = torch.empty(rows, widths[L], dtype=torch.bfloat16, device=dev)
    wl = _i32_array(widths)
    if packed_ws is not None:      # fused_mlp_pack(..., family=fam) ran before (the caller ordered the two)
        ws, ws_bytes, phase = packed_ws, packed_ws.numel(), MLP_PHASE_RUN
    else:
        ws_bytes = size_query("trs_mlp_fused_workspace_bytes", L, wl)
        ws, phase = torch.empty(ws_bytes, dtype=torch.uint8, device=dev), MLP_PHASE_ALL
    call("trs_mlp_fused_fwd", ptr(x2), rows, L, wl, _ptr_array(Ws), _ptr_array(bs), _ptr_array(hidden),
         _ptr_array(masks), ptr(mask_in), ptr(y), _abi.TRS_BF16, fam, phase, int(x_stride), ptr(ws), ws_bytes, stream_ptr())
    if input_mask:
        return y, hidden, masks, mask_in, fam
    return y, hidden, masks, fam


def fused_mlp_backward_raw(gy2: torch.Tensor, widths: Sequence[int], Ws: Sequence[torch.Tensor],
                           masks: Sequence[torch.Tensor], mask_in: Optional[torch.Tensor] = None, *, family: int,
                           packed_ws: Optional[torch.Tensor] = None):
    """trs_mlp_fused_bwd_data: (gx, gz [d(pre-activation) of the hidden layers], gb [fp32 bias gradients, padded]) and,
    with ``mask_in``, gb_in: gx is then masked by the upstream ReLU and gb_in holds its column sums.  ``family``: what
    fused_mlp_forward_raw returned with these masks."""
    if family not in (MLP_FAMILY_TILE, MLP_FAMILY_ROW_OWNER, MLP_FAMILY_MIXED):
        raise ValueError("fused_mlp_backward_raw: family must be the value the forward returned with these masks")
    L = len(Ws)
    rows, dev = gy2.shape[0], gy2.device
    gz = [torch.empty(rows, _pad32(widths[l + 1]), dtype=torch.bfloat16, device=dev) for l in range(L - 1)]
    gb = [torch.empty(_pad32(widths[l + 1]), dtype=torch.float32, device=dev) for l in range(L)]
    gx = torch.empty(rows, widths[0], dtype=torch.bfloat16, device=dev)   # the kernel always writes dL/dx (its last GEMM)
    gb_in = torch.empty(_pad32(widths[0]), dtype=torch.float32, device=dev) if mask_in is not None else None
    wl = _i32_array(widths)
    if packed_ws is not None:
        ws, ws_bytes, phase = packed_ws, packed_ws.numel(), MLP_PHASE_RUN
    else:
        ws_bytes = size_query("trs_mlp_fused_workspace_bytes", L, wl)
        ws, phase = torch.empty(ws_bytes, dtype=torch.uint8, device=dev), MLP_PHASE_ALL
    call("trs_mlp_fused_bwd_data", ptr(gy2), rows, L, wl, _ptr_array(Ws), _ptr_array(masks), _ptr_array(gz),
         _ptr_array(gb), ptr(gx), ptr(mask_in), ptr(gb_in), _abi.TRS_BF16, int(family), phase, ptr(ws), ws_bytes,
         stream_ptr())
    return gx, gz, gb, gb_in


class _FusedMLP(Function):
    """y = Linear_{L-1}(relu(... relu(Linear_0(x)))) on the rows of x (..., widths[0]); params = W0, b0, W1, b1, ...
    (nn.Linear layout).  Forward and the data gradient are one HIP kernel each (trs_mlp_fused_*); the weight gradients
    are GEMMs with K = rows on the tensors those kernels leave behind (hidden activations, pre-activation gradients)."""

    @staticmethod
    def forward(ctx, x, *params):
        require_device(x, *params)
        L = len(params) // 2
        Ws = [params[2 * l].contiguous() for l in range(L)]
        bs = [params[2 * l + 1].contiguous() for l in range(L)]
        widths = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
        x2 = x.reshape(-1, widths[0]).contiguous()
        y, hidden, masks, fam = fused_mlp_forward_raw(x2, Ws, bs)
        ctx.save_for_backward(x2, *Ws, *hidden, *masks)
        ctx.meta = (L, widths, tuple(x.shape), [p.dtype for p in params], fam)
        return y.reshape(*x.shape[:-1], widths[L])

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        L, widths, xshape, pdt, fam = ctx.meta
        saved = ctx.saved_tensors
        x2, Ws = saved[0], saved[1:1 + L]
        hidden, masks = saved[1 + L:L + L], saved[L + L:]
        rows, dev = x2.shape[0], x2.device
        gy2 = gy.reshape(rows, widths[L]).contiguous()
        gx, gz, gb, _ = fused_mlp_backward_raw(gy2, widths, Ws, masks, family=fam)
        grads = []
        for l in range(L):
            inp = x2 if l == 0 else hidden[l - 1]               # (rows, widths[l] | pad32)
            g = gy2 if l == L - 1 else gz[l]                    # (rows, widths[l+1] | pad32)
            gw = gbias = None
            need_w, need_b = ctx.needs_input_grad[1 + 2 * l], ctx.needs_input_grad[2 + 2 * l]
            if need_w and need_b and pdt[2 * l] == pdt[2 * l + 1]:
                gw, gbias = _wgrad_rows(g, inp, widths[l + 1], widths[l], pdt[2 * l], gb[l])
            else:
                if need_w:
                    gw = _wgrad_rows(g, inp, widths[l + 1], widths[l], pdt[2 * l])
                if need_b:
                    gbias = gb[l][:widths[l + 1]].to(pdt[2 * l + 1])
            grads += [gw, gbias]
        return (gx.reshape(xshape) if ctx.needs_input_grad[0] else None, *grads)


ROWS_GEMM = os.environ.get("TRS_ROWS_GEMM", "1") not in ("", "0")
ROWS_GEMM_MIN_COLS = 1024       # narrower outputs: the library GEMM is as fast


def rows_gemm_supported_for(rows: int, x_stride: int, W: torch.Tensor, out_f: int, in_f: int) -> bool:
    """rows_gemm_supported for a contiguous bf16 (rows, x_stride) gradient that does not exist yet"""
    if not (ROWS_GEMM and W.is_cuda and W.dtype == torch.bfloat16 and W.is_contiguous() and rows >= 4096
            and in_f >= ROWS_GEMM_MIN_COLS and W.shape[1] == in_f and W.shape[0] >= out_f):
        return False
    return bool(_abi.load().trs_rows_gemm_supported(int(out_f), int(in_f), int(x_stride)))


def rows_gemm_supported(g: torch.Tensor, W: torch.Tensor, out_f: int, in_f: int) -> bool:
    """dL/dx = g[:, :out_f] @ W[:out_f] by trs_rows_gemm: bf16 on the device, a short contraction (out_f <= 512) and a
    wide result (the 2496 embedding columns in front of a deep branch), enough rows to fill the chip"""
    if not (g.is_cuda and g.dtype == torch.bfloat16 and g.dim() == 2 and g.is_contiguous()):
        return False
    return rows_gemm_supported_for(g.shape[0], g.shape[1], W, out_f, in_f)


def rows_gemm_pack(W: torch.Tensor, rows: int, x_stride: int, out_f: int, in_f: int) -> torch.Tensor:
    """The PACK phase of trs_rows_gemm on the current stream (see fused_mlp_pack): W in fragment order in a fresh
    workspace for ``rows_gemm(..., packed_ws=...)``"""
    ws_bytes = size_query("trs_rows_gemm_workspace_bytes", out_f, in_f)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=W.device)
    call("trs_rows_gemm", ptr(None), rows, x_stride, ptr(W), out_f, in_f, _abi.TRS_BF16, MLP_PHASE_PACK, ptr(None), ptr(ws),
         ws_bytes, stream_ptr())
    return ws


def rows_gemm(g: torch.Tensor, W: torch.Tensor, out_f: int, in_f: int,
              packed_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    y = torch.empty(g.shape[0], in_f, dtype=torch.bfloat16, device=g.device)
    if packed_ws is not None:
        ws, ws_bytes, phase = packed_ws, packed_ws.numel(), MLP_PHASE_RUN
    else:
        ws_bytes = size_query("trs_rows_gemm_workspace_bytes", out_f, in_f)
        ws, phase = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device), MLP_PHASE_ALL
    call("trs_rows_gemm", ptr(g), g.shape[0], g.shape[1], ptr(W), out_f, in_f, _abi.TRS_BF16, phase, ptr(y), ptr(ws),
         ws_bytes, stream_ptr())
    return y


def _tail_layer_grads(g, inp, out_f, in_f, dtype, gb_f32, need_w, need_b, splits_div=1):
    """(dW, db) of one layer behind the fused kernels: the bias cast rides in the weight gradient's finish launch"""
    if need_w and need_b:
        return _wgrad_rows(g, inp, out_f, in_f, dtype, gb_f32, splits_div=splits_div)
    gw = _wgrad_rows(g, inp, out_f, in_f, dtype, splits_div=splits_div) if need_w else None
    return gw, (gb_f32[:out_f].to(dtype) if need_b else None)


class _FusedMLPTail(Function):
    """The layers BEHIND a wide first layer (DeepFM / xDeepFM deep branch: 2496 -> 400 | -> 400 -> 400 -> 1) as one HIP
    kernel per direction: x2 is the first layer's ReLU output at its zero-padded GEMM width, the tail's first weight is
    padded on the input side to match and its last on the output side to the kernel's minimum width (8 columns; only the
    first ``out_f`` are returned).  ``tensors`` per layer: weight, bias (the parameters: they receive the gradients),
    w_use, b_use (what the kernel reads: the parameter itself when None)."""

    @staticmethod
    def forward(ctx, x2, *tensors):
        require_device(x2, *[t for t in tensors if t is not None])
        L = len(tensors) // 4
        Ws = [(tensors[4 * l] if tensors[4 * l + 2] is None else tensors[4 * l + 2]).contiguous() for l in range(L)]
        bs = [(tensors[4 * l + 1] if tensors[4 * l + 3] is None else tensors[4 * l + 3]).contiguous() for l in range(L)]
        y, hidden, masks, fam = fused_mlp_forward_raw(x2, Ws, bs)
        out_f = tensors[4 * (L - 1)].shape[0]
        ctx.save_for_backward(x2, *Ws, *hidden, *masks)
        ctx.meta = (L, [x2.shape[1]] + [w.shape[0] for w in Ws], [tuple(tensors[4 * l].shape) for l in range(L)],
                    [tensors[4 * l].dtype for l in range(L)], fam)
        return y[:, :out_f] if out_f != y.shape[1] else y      # (a view of the padded output: consumers read it strided)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        L, widths, wshapes, wdt, fam = ctx.meta
        saved = ctx.saved_tensors
        x2, Ws = saved[0], saved[1:1 + L]
        hidden, masks = saved[1 + L:L + L], saved[L + L:]
        rows, dev = x2.shape[0], x2.device
        gy2 = pad_cols(gy, widths[L]) if gy.shape[1] != widths[L] else gy.contiguous()
        gx, gz, gb, _ = fused_mlp_backward_raw(gy2, widths, Ws, masks, family=fam)
        grads = []
        for l in range(L):
            inp = x2 if l == 0 else hidden[l - 1]
            g = gy2 if l == L - 1 else gz[l]
            out_f, in_f = wshapes[l]
            gw, gbias = _tail_layer_grads(g, inp, out_f, in_f, wdt[l], gb[l], ctx.needs_input_grad[1 + 4 * l],
                                          ctx.needs_input_grad[2 + 4 * l])
            grads += [gw, gbias, None, None]
        return (gx if ctx.needs_input_grad[0] else None, *grads)


WGRAD_ROWS = os.environ.get("TRS_WGRAD_ROWS", "1") not in ("", "0")


def pad_cols(g: torch.Tensor, width: int) -> torch.Tensor:
    """(rows, c) bf16 -> (rows, width) with zeros behind the c columns, one launch (trs_pad_cols)"""
    g = g.contiguous()
    out = torch.empty(g.shape[0], width, dtype=g.dtype, device=g.device)
    call("trs_pad_cols", ptr(g), g.shape[1], ptr(out), width, g.shape[0], value_dtype_code(g), stream_ptr())
    return out


def _wgrad_rows(g: torch.Tensor, inp: torch.Tensor, out_f: int, in_f: int, dtype, gb_f32: Optional[torch.Tensor] = None,
                splits_div: int = 1):
    """dW = g^T @ inp over the rows (K = rows): split-K batched GEMM with fp32 partials, folded / sliced / cast by
    trs_wgrad_finish (the padding columns of g / inp are dropped there).  With ``gb_f32`` (the layer's fp32 bias
    gradient, at least out_f entries) returns (dW, db): the cast of the bias gradient rides in the same finish launch."""
    if gb_f32 is not None:
        gb = torch.empty(out_f, dtype=dtype, device=g.device)
        if out_f <= (in_f + 255) // 256 * 256:
            return _wgrad_rows_impl(g, inp, out_f, in_f, dtype, gb_f32, gb, splits_div), gb
        gb.copy_(gb_f32[:out_f])
        return _wgrad_rows_impl(g, inp, out_f, in_f, dtype, None, None, splits_div), gb
    return _wgrad_rows_impl(g, inp, out_f, in_f, dtype, None, None, splits_div)


def wgrad_rows_splits(g: torch.Tensor, inp: torch.Tensor, out_f: int, in_f: int) -> int:
    """row ranges trs_wgrad_rows cuts this layer's weight gradient into (0: the kernel does not take the shape)"""
    if not (WGRAD_ROWS and g.is_cuda and g.dtype == torch.bfloat16 and inp.dtype == torch.bfloat16 and g.is_contiguous()
            and inp.is_contiguous()):
        return 0
    M, N = min(g.shape[1], (out_f + 7) // 8 * 8), min(inp.shape[1], (in_f + 7) // 8 * 8)
    return int(_abi.load().trs_wgrad_rows_splits(M, N, int(g.shape[0])))


def _wgrad_rows_impl(g, inp, out_f, in_f, dtype, gb_f32, gb, splits_div=1):
    rows = g.shape[0]
    if (WGRAD_ROWS and g.is_cuda and g.dtype == torch.bfloat16 and inp.dtype == torch.bfloat16 and g.is_contiguous()
            and inp.is_contiguous()):
        # only the columns that exist in the weight (rounded up to the kernel's 8): a 400-wide layer kept in 416-column
        # tensors is 25 x 25 output tiles instead of 26 x 26, and its 16 zero columns are not fetched
        M, N = min(g.shape[1], (out_f + 7) // 8 * 8), min(inp.shape[1], (in_f + 7) // 8 * 8)
        S = int(_abi.load().trs_wgrad_rows_splits(M, N, int(rows)))
        if S > 0:
            while splits_div > 1 and S % 2 == 0 and S // 2 >= 8:      # (S is 8 x a power of two)
                S //= 2
                splits_div //= 2
            part = torch.empty(S, M, N, dtype=torch.float32, device=g.device)
            call("trs_wgrad_rows", ptr(g), g.shape[1], ptr(inp), inp.shape[1], rows, M, N,
                 _abi.TRS_BF16, S, ptr(part), stream_ptr())
            gw = torch.empty(out_f, in_f, dtype=dtype, device=g.device)
            call("trs_wgrad_finish", ptr(part), S, M, N, out_f, in_f, value_dtype_code(gw),
                 ptr(gw), ptr(gb_f32), ptr(gb), stream_ptr())
            return gw
    S = 0
    # 192 batches measured best at 2.5 M rows (416 x 416: 1.33 ms against 1.48 at 384 and 1.53 at 96; 416 x 64: 0.49 against
    # 0.57 / 0.54): enough workgroups for the chip, partials still small against the operands
    for cand in (192, 128, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4):
        if rows % cand == 0 and rows // cand >= 1024:
            S = cand
            break
    if S == 0:
        if gb is not None:
            gb.copy_(gb_f32[:out_f])
        return (g.t() @ inp)[:out_f, :in_f].contiguous().to(dtype)
    part = torch.bmm(g.view(S, rows // S, -1).transpose(1, 2), inp.view(S, rows // S, -1), out_dtype=torch.float32)
    gw = torch.empty(out_f, in_f, dtype=dtype, device=g.device)
    call("trs_wgrad_finish", ptr(part), S, part.shape[1], part.shape[2], out_f, in_f, value_dtype_code(gw), ptr(gw),
         ptr(gb_f32), ptr(gb), stream_ptr())
    return gw


def fused_mlp(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return _FusedMLP.apply(x, *params)


# --------------------------------------------------------------------------------------------
# K8: the scalar head of the CTR models and its loss (csrc/head.hip)
# --------------------------------------------------------------------------------------------
class _CTRLogit(Function):
    """logit (B,1) = sum_e fm + sum_n feat + sum_k extra_k + bias in one pass; the gradient of every operand is the
    incoming column broadcast, returned as VIEWS (the lookups' backwards read a broadcast gradient as one value per
    sample: _fm_grad_operand, _GatherRows.backward) -- no kernel in the backward except the bias' batch sum."""

    @staticmethod
    def forward(ctx, fm, feat, bias, *extras):
        ref = fm if fm is not None else (feat if feat is not None else extras[0])
        require_device(ref, *[t for t in (fm, feat, bias, *extras) if t is not None])
        B = ref.shape[0]
        dt = ref.dtype
        for t in (fm, feat, bias, *extras):
            if t is not None and t.dtype != dt:
                raise ValueError("ctr_logit: every operand must have the same dtype")
        fmc = None if fm is None else fm.contiguous()
        ftc = None if feat is None else feat.reshape(B, -1).contiguous()
        E = 0 if fmc is None else fmc.shape[1]
        N = 0 if ftc is None else ftc.shape[1]
        cols, strides = [], []
        for t in extras:
            if t.dim() != 2 or t.shape[0] != B or t.shape[1] != 1:
                raise ValueError(f"ctr_logit: extra terms must be (B,1), got {tuple(t.shape)}")
            cols.append(t)
            strides.append(t.stride(0))          # a (B,1) slice of a wider row (the padded logit column of the fused MLP tail)
        out = torch.empty(B, 1, dtype=dt, device=ref.device)
        call("trs_ctr_logit_fwd", ptr(fmc), E, ptr(ftc), N, _ptr_array(cols), (ctypes.c_int64 * max(1, len(cols)))(*strides),
             len(cols), ptr(bias), B, value_dtype_code(out), ptr(out), stream_ptr())
        ctx.shapes = (None if fm is None else tuple(fm.shape), None if feat is None else tuple(feat.shape),
                      None if bias is None else tuple(bias.shape), len(extras))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        fm_shape, feat_shape, bias_shape, n_extras = ctx.shapes
        needs = ctx.needs_input_grad
        B = g.shape[0]
        g_fm = g.expand(fm_shape) if fm_shape is not None and needs[0] else None
        g_feat = None
        if feat_shape is not None and needs[1]:
            g_feat = g.reshape(B, *([1] * (len(feat_shape) - 1))).expand(feat_shape)
        g_bias = g.sum().reshape(bias_shape) if bias_shape is not None and needs[2] else None
        return (g_fm, g_feat, g_bias, *[g if needs[3 + k] else None for k in range(n_extras)])


def ctr_logit(fm: Optional[torch.Tensor] = None, feat: Optional[torch.Tensor] = None,
              extras: Sequence[torch.Tensor] = (), bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The scalar head the reference's CTR models end in, as one kernel: ``fm`` (B,E) second-order FM term (summed over
    E), ``feat`` (B,N,1) | (B,N) first-order weights of the looked-up rows (summed over N), ``extras`` up to four (B,1)
    columns (deep / CIN outputs), ``bias`` one value -> (B,1).  models/ctr/factorization_machine.py:55-66,
    deep_fm.py:75-104, xdeep_fm.py:117-121."""
    if fm is None and feat is None and not extras:
        raise ValueError("ctr_logit: nothing to sum")
    strip = lambda t: None if t is None else (t.rename(None) if t.has_names() else t)
    return _CTRLogit.apply(strip(fm), strip(feat), bias, *[strip(t) for t in extras])


class _BCEWithLogits(Function):
    @staticmethod
    def forward(ctx, logits, labels):
        require_device(logits, labels)
        x = logits.reshape(-1).contiguous()
        y = labels.reshape(-1).contiguous()
        if x.numel() != y.numel():
            raise ValueError(f"bce_with_logits: {tuple(logits.shape)} logits against {tuple(labels.shape)} labels")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        ws_bytes = size_query("trs_bce_logits_workspace_bytes", x.numel())
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        call("trs_bce_logits_fwd", ptr(x), value_dtype_code(x), ptr(y), value_dtype_code(y), x.numel(), ptr(loss), ptr(ws),
             ws_bytes, stream_ptr())
        ctx.save_for_backward(x, y)
        ctx.shape = tuple(logits.shape)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = torch.empty_like(x)
        gf = g.float().contiguous()
        call("trs_bce_logits_bwd", ptr(x), value_dtype_code(x), ptr(y), value_dtype_code(y), ptr(gf), x.numel(), ptr(gx),
             stream_ptr())
        return gx.reshape(ctx.shape), None


def bce_with_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """mean binary cross entropy of ``logits`` (fp32 / bf16, any shape) against ``labels`` (fp32 / bf16, same number of
    elements): an fp32 scalar.  ``F.binary_cross_entropy_with_logits(logits.float(), labels)`` in three launches
    (forward 2, backward 1) instead of ATen's cast + log-sigmoid chain + mean and their backwards."""
    return _BCEWithLogits.apply(logits, labels)
