"""Row-sharded embedding table over the GPUs of one node (one process per GPU, torch.distributed).

New design (the reference has no distributed code at all -- SURVEY.md section 2.3 / 8e): the concatenated
``sum(field_sizes) x E`` table of ``MultiIndicesEmbedding`` (multi_indices_emb.py:48) is split ROW-wise
into ``world`` contiguous ranges; every rank keeps its local batch (data parallel) and

  forward   1. bucket the local (B*N) global row ids by owner rank           (HIP: trs_bucket_by_owner)
            2. all-to-all the per-owner counts, then the local row ids        (int32 on the wire)
            3. owners gather the requested rows from their shard              (HIP: trs_gather_rows)
            4. all-to-all the rows back                                        (bf16 / fp32 on the wire)
            5. un-permute into the (B,N,E) block -- fused with the FM second-order term when asked
               (HIP: trs_embed_fm with the received rows as the "table" and the inverse permutation
               as the index)
  backward  the block gradient is permuted into exchange order, sent back by the reverse all-to-all and
            reduced into the owner's shard: a sparse reduce-scatter keyed by row id.  The shard gradient is
            dense for small shards and an (uncoalesced) sparse COO tensor otherwise -- at 125 M rows per
            GPU a dense gradient would be 16 GB per step (documented divergence from nn.Embedding's
            sparse=False default).

With ``set_fused_optimizer(opt)`` the owner applies the optimizer step to the rows it received gradients for inside
the backward pass (no gradient tensor at all: the mandatory mode for 125 M-row shards).
``dedup=True`` sends every distinct row id of the local batch once (torch.unique before the exchange): the block is
rebuilt from the distinct rows and the gradients of duplicate lookups are summed before they travel -- pays on skewed
(Zipf) data, costs a sort on uniform data, default off.  A process group of one rank takes no collective and no host
read at all (the step is then capturable into a hipGraph).

``capacity=<factor>`` switches the exchange to FIXED-CAPACITY slots: every peer gets ``ceil(B*N/world * factor)`` lookup
slots per step (padded with -1), so all three all-to-alls have equal, compile-time splits and NO split size is ever read on
the host -- the host can run ahead of the device and the whole step (collectives included) has a static launch sequence,
which is what a hipGraph capture needs.  A peer whose share exceeds its slots raises the device-side index flag
(``functional.index_errors_seen()``) and the lookups that did not fit read as zero rows; with uniform ids a factor of
1.1 is ~100 standard deviations of the per-peer count at the BASELINE shape.

Cross-step pipeline (round 4; the 8-GPU step is link-bound, so the exchanges have to hide behind the dense part):
``prefetch_lookup(next_inputs)`` runs steps 1-4 of the NEXT batch -- route, id all-to-all, owner-side gather, row
all-to-all -- on a communication stream while the current batch's MLP forward / backward occupies the compute stream; the
next ``forward`` finds the received rows ready and only un-permutes them.  ``overlap_grad_exchange=True`` moves the reverse
all-to-all and the owner-side reduction of a batch onto the same stream, where they run under the NEXT batch's forward
(``wait_grad()`` makes the current stream wait for them; ``weight.grad`` must not be READ before).  With gradients only
(the fwd+bwd metric) nothing is stale: the shard does not change between the early lookup and the forward that uses it.
With a fused optimizer on the owner an early lookup reads rows that are one update behind (asynchronous-SGD semantics):
refused unless ``prefetch_lookup(..., stale_ok=True)``.

``backend='nccl'`` is RCCL on ROCm (xGMI links); the same code runs on ``gloo`` with CPU tensors when a
CPU ``ops`` object is injected (tests only -- the default ops are the HIP kernels and refuse CPU tensors).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import functional as F_
from ._abi import call, index_dtype_code, ptr, require_device, size_query, stream_ptr, value_dtype_code
from .inputs import BaseInput, field_offsets

DENSE_GRAD_MAX_ROWS = 8_000_000      # shards up to this many rows get a dense gradient
# TRS_SHARD_PREFETCH: build the owner-side row buckets at forward time on the side stream (1), never (0), or -- default --
# only in the pipelined step (overlap_grad_exchange), where the backward's owner reduction runs on the communication stream
# under the next forward and finds them ready.  Measured on the one-rank DeepFM step, same box, alternating runs (round 5,
# profiles/r05_logs/ab_shard_prefetch.txt): pipelined 1.787 -> 1.725 ms with it; un-pipelined 1.712 -> 1.758 ms (there the
# global-atomic build -- no per-field ranges on the owner side -- slows the bandwidth-bound lookup kernels it runs beside by
# more than the ~170 us it takes off the backward).
OWNER_PREFETCH = __import__("os").environ.get("TRS_SHARD_PREFETCH", "auto")


class HipOps:
    """Device-side pieces of the sharded lookup, on libtrs_hip.so."""
    _uniq_cache = None      # (key, ids kept alive, distinct ids, inverse) of the last compact-row optimizer step

    def bucket_by_owner(self, idx: torch.Tensor, offsets: torch.Tensor, rows_per_rank: int, world: int):
        require_device(idx, offsets)
        B, N = idx.shape
        BN = B * N
        dev = idx.device
        counts = torch.empty(world, dtype=torch.int64, device=dev)
        send_ids = torch.empty(BN, dtype=torch.int32, device=dev)
        send_pos = torch.empty(BN, dtype=torch.int32, device=dev)
        inv_pos = torch.empty(BN, dtype=torch.int32, device=dev)
        ws_bytes = size_query("trs_bucket_workspace_bytes", BN, world)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        call("trs_bucket_by_owner", ptr(idx), index_dtype_code(idx), ptr(offsets), B, N, rows_per_rank, world,
             ptr(counts), ptr(send_ids), ptr(send_pos), ptr(inv_pos), ptr(ws), ws_bytes, stream_ptr())
        return counts, send_ids, send_pos, inv_pos

    def gather_local(self, weight: torch.Tensor, ids: torch.Tensor, n_valid: Optional[int] = None,
                     padded: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rows ``ids`` of this rank's shard.  Ids outside [0, n_valid) -- a global id past the table, a negative id,
        the short last shard -- read as zero rows and raise the device-side index flag
        (functional.index_errors_seen() / TRS_CHECK_INDICES=1) instead of touching foreign memory.
        ``padded``: -1 entries are the padding of a fixed-capacity exchange (their rows are never used): they read row 0
        and raise nothing (negative REAL ids were flagged by the sender, ``pad_slots``)."""
        if padded:
            ids = ids.clamp_min(0)
        K = ids.numel()
        V, E = weight.shape
        if out is None:
            out = torch.empty(K, E, dtype=weight.dtype, device=weight.device)
        if K:
            flag = F_._ErrFlag(weight.device)
            call("trs_gather_rows", ptr(weight), V if n_valid is None else min(V, n_valid), E, value_dtype_code(weight),
                 ptr(ids), index_dtype_code(ids), ptr(None), K, 1, ptr(out), ptr(flag.t), stream_ptr())
            flag.check("sharded lookup")
        return out

    def unique_route(self, idx: torch.Tensor, offsets: torch.Tensor, rows_per_rank: int, world: int):
        """De-duplicated routing: (counts per owner (W int64), LOCAL ids of the distinct rows grouped by owner (U int32),
        slot of every lookup in that list (B*N int32))."""
        require_device(idx, offsets)
        g = (idx.long() + offsets.view(1, -1)).reshape(-1)
        uniq, inv = torch.unique(g, return_inverse=True)            # ascending global ids: grouped by owner already
        owner = torch.div(uniq, rows_per_rank, rounding_mode="floor").clamp_(0, world - 1)
        counts = torch.bincount(owner, minlength=world).to(torch.int64)
        return counts, (uniq - owner * rows_per_rank).to(torch.int32), inv.to(torch.int32)

    def reduce_grad_unique(self, g_block, inv, rows, g_fm, fm_sum):
        """gradient of the U distinct rows: sum over the lookups that point at each (+ the folded FM term, with the
        received rows standing in for the table) -- the row-bucket reduction of the unsharded backward on a U-row space."""
        U, E = rows.shape
        K = inv.numel()
        rb = F_.row_buckets(inv.view(K, 1), None, U)
        gb = None if g_block is None else g_block.reshape(K, 1, E).contiguous()
        if g_fm is not None:
            # lookups per sample: the FM operands are indexed by sample = lookup // N
            rb = F_.RowBuckets(rb.row_start, rb.perm, rb.V, rb.BN, K // g_fm.shape[0])
            return F_.scatter_rows(rb, rows, g_rows=gb, g_bcast=F_._fm_grad_operand(g_fm), fm_sum=fm_sum)
        return F_.scatter_rows(rb, rows, g_rows=gb)

    def shard_update(self, weight: torch.Tensor, ids: torch.Tensor, grad_rows: torch.Tensor, opt, dense_index: bool,
                     padded: bool = False):
        """fused optimizer step on the owner: rows ``ids`` (with repeats) of ``weight`` receive ``grad_rows``
        (``padded``: -1 entries are padding slots and update nothing)"""
        if ids.numel() == 0:          # this rank received no lookups this step
            return
        with torch.no_grad():
            if dense_index:
                rb = F_.row_buckets(ids.view(-1, 1), None, weight.shape[0], check=not padded)
                F_.scatter_rows_update(rb, weight.data, opt, g_rows=grad_rows.contiguous(), key=weight)
            else:
                # the distinct touched rows of this step: shared by every table looked up with the same indices (the
                # E = 64 table and its E = 1 companion receive the same owner ids: one sort instead of two)
                key = (ids.data_ptr(), ids._version, ids.numel())
                cur = torch.cuda.current_stream(ids.device)
                hit = HipOps._uniq_cache if HipOps._uniq_cache is not None and HipOps._uniq_cache[0] == key else None
                if hit is None:
                    uniq, inv = torch.unique(ids, return_inverse=True)
                    inv32 = inv.to(torch.int32).view(-1, 1)
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    HipOps._uniq_cache = hit = (key, ids, uniq, inv32, ev, cur)
                _, _, uniq, inv32, ev, made_on = hit
                if cur != made_on:        # the companion table's backward may run on the "lookup" side stream
                    cur.wait_event(ev)
                    uniq.record_stream(cur)
                    inv32.record_stream(cur)
                rb = F_.row_buckets(inv32, None, uniq.numel())
                # ids outside the shard (the lookup read them as zero rows and raised the index flag) stay in ``uniq``:
                # trs_scatter_rows_update_mapped skips row_map entries outside [0, V), so they update nothing
                F_.scatter_rows_update_mapped(rb, weight.data, opt, grad_rows, uniq.to(torch.int32), key=weight)

    def prefetch_owner_buckets(self, weight: torch.Tensor, ids: torch.Tensor, padded: bool = False,
                               pipelined: bool = False) -> None:
        """The owner-side row buckets of this step (the CSR over the ids this rank RECEIVED) depend only on the route: start
        building them on the side stream at forward time, as the unsharded lookup does, so the backward's reduction
        finds them ready instead of running a ~160 us global-atomic build on the critical path."""
        on = OWNER_PREFETCH == "1" or (OWNER_PREFETCH == "auto" and pipelined)
        if ids.numel() and weight.requires_grad and on:
            F_.prefetch_row_buckets(ids.view(-1, 1), None, weight.shape[0], check=not padded)

    def unpermute(self, rows: torch.Tensor, inv_pos: torch.Tensor, B: int, N: int, want_fm: bool, out=None):
        """block[p] = rows[inv_pos[p]]; with ``want_fm`` also FM second order + the fp32 field sum.  ``out``: (block, fm,
        fm_sum) buffers to write into (the module's persistent output buffers) instead of fresh allocations."""
        K, E = rows.shape
        fm = fm_sum = None
        if out is not None:
            block, fm, fm_sum = out
        else:
            block = torch.empty(B, N, E, dtype=rows.dtype, device=rows.device)
            if want_fm:
                fm = torch.empty(B, E, dtype=rows.dtype, device=rows.device)
                fm_sum = torch.empty(B, E, dtype=torch.float32, device=rows.device)
        if B:
            call("trs_embed_fm", ptr(rows), max(K, 1), E, value_dtype_code(rows), ptr(inv_pos), index_dtype_code(inv_pos),
                 ptr(None), B, N, ptr(block), ptr(fm), ptr(fm_sum), ptr(None), ptr(None), ptr(None), stream_ptr())
        return block, fm, fm_sum

    def unpermute_local(self, back: torch.Tensor, weight: torch.Tensor, n_valid: int, inv_pos: torch.Tensor,
                        send_ids: torch.Tensor, self_lo: int, self_n: int, B: int, N: int, want_fm: bool, out=None):
        """``unpermute`` with the lookups this rank owns itself -- slots [self_lo, self_lo + self_n) of the exchange order --
        read straight from its shard (trs_embed_fm_sharded); ``back`` holds the received rows WITHOUT that segment."""
        E = weight.shape[1]
        fm = fm_sum = None
        if out is not None:
            block, fm, fm_sum = out
        else:
            block = torch.empty(B, N, E, dtype=weight.dtype, device=weight.device)
            if want_fm:
                fm = torch.empty(B, E, dtype=weight.dtype, device=weight.device)
                fm_sum = torch.empty(B, E, dtype=torch.float32, device=weight.device)
        if B:
            flag = F_._ErrFlag(weight.device)
            call("trs_embed_fm_sharded", ptr(back if back.numel() else None), back.shape[0], ptr(weight), weight.shape[0],
                 min(weight.shape[0], n_valid), E, value_dtype_code(weight), ptr(inv_pos), ptr(send_ids), int(self_lo),
                 int(self_n), B, N, ptr(block), ptr(fm), ptr(fm_sum), ptr(flag.t), stream_ptr())
            flag.check("sharded lookup")
        return block, fm, fm_sum

    def permute_grad(self, g_block: Optional[torch.Tensor], send_pos: torch.Tensor, g_fm, fm_sum, block, out=None):
        """rows of d(block) in exchange order: g_block[pos[k]] (+ g_fm*(S - x) when the FM term was fused) -- one
        kernel (trs_permute_grad).  ``out``: (K, E) rows to write into."""
        B, N, E = block.shape
        K = send_pos.numel()
        gb = None if g_block is None else g_block.contiguous()
        gf = None if g_fm is None else g_fm.contiguous()
        if out is None:
            out = torch.empty(K, E, dtype=block.dtype, device=block.device)
        if K:
            call("trs_permute_grad", ptr(gb), ptr(gf), ptr(fm_sum if gf is not None else None),
                 ptr(block if gf is not None else None), ptr(send_pos), K, N, E, value_dtype_code(block), ptr(out),
                 stream_ptr())
        return out

    # ---- dense shard gradient with the lookups this rank owns itself reduced STRAIGHT from the block gradient: the
    # unsharded backward (functional._EmbedFM.backward) on this rank's rows -- per-field LDS-counter bucket build started at
    # forward time on the side stream, one bucket walk with the FM term folded in -- instead of permute (1 GB of traffic at
    # the BASELINE shape) + a global-atomic bucket build over ids without field structure + a plain walk.
    def prefetch_own_buckets(self, weight: torch.Tensor, idx: torch.Tensor, offsets_local: torch.Tensor) -> None:
        if weight.requires_grad:
            F_.prefetch_row_buckets(idx, offsets_local, weight.shape[0], check=False)

    def own_grad_dense(self, weight: torch.Tensor, idx: torch.Tensor, offsets_local: torch.Tensor, g_block, g_fm,
                       fm_sum) -> torch.Tensor:
        """``offsets_local`` = field offsets minus the first row of the shard: a lookup of another rank's row lands
        outside [0, rows of the shard) and is skipped (check=False: no index flag for those)"""
        rb = F_.row_buckets(idx, offsets_local, weight.shape[0], check=False)
        if g_fm is not None:
            return F_.scatter_rows(rb, weight, g_rows=None if g_block is None else g_block.contiguous(),
                                   g_bcast=F_._fm_grad_operand(g_fm), fm_sum=fm_sum)
        return F_.scatter_rows(rb, weight, g_rows=g_block.contiguous())

    def accumulate_rows(self, gw: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor, padded: bool = False) -> None:
        """gw[ids[k]] += rows[k] (fp32 sums per row, one read-modify-write per touched row): the gradient rows that
        arrived over the wire, added to the dense gradient of this rank's own lookups"""
        if ids.numel() == 0:
            return
        rb = F_.row_buckets(ids.view(-1, 1), None, gw.shape[0], check=not padded)
        with torch.no_grad():
            F_.scatter_rows_update(rb, gw, _ACCUMULATE, g_rows=rows.contiguous(), key=gw)

    def shard_grad_dense(self, weight: torch.Tensor, ids: torch.Tensor, grad_rows: torch.Tensor,
                         padded: bool = False) -> torch.Tensor:
        rb = F_.row_buckets(ids.view(-1, 1), None, weight.shape[0], check=not padded)
        return F_.scatter_rows(rb, weight, g_rows=grad_rows.contiguous())


class _Accumulate:
    """the fused-SGD row sink with lr = -1: w[r] -= lr * g[r] adds the reduced rows into the table it is pointed at"""
    kind, lr, eps = 1, -1.0, 0.0

    def state_for(self, table, key=None):
        return None


_ACCUMULATE = _Accumulate()


def shard_ranges(num_rows: int, world: int):
    per = (num_rows + world - 1) // world
    return per, [(min(num_rows, r * per), min(num_rows, (r + 1) * per)) for r in range(world)]


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int], group):
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


PAD = -1        # id / position of a padding slot
# TRS_SHARD_FORCE_COLLECTIVES=1 (tests): a one-rank group normally takes no collective at all; with this switch it issues
# the same all_to_all_single calls a larger world does (RCCL sending to itself), so that their capture into a hipGraph can
# be exercised on the one GPU the build boxes have
FORCE_COLLECTIVES = __import__("os").environ.get("TRS_SHARD_FORCE_COLLECTIVES", "0") == "1"

# ---- communication stream and per-phase device times --------------------------------------------------------------------
_comm_streams = {}


def comm_stream(dev: torch.device):
    """The stream the exchanges of the pipelined step run on (one per device); None for CPU tensors (gloo tests: the same
    code path runs inline, in program order)."""
    if dev.type != "cuda":
        return None
    s_ = _comm_streams.get(dev)
    if s_ is None:
        s_ = _comm_streams[dev] = torch.cuda.Stream(device=dev)
    return s_


class _on_stream:
    """``with _on_stream(s):`` -- make ``s`` current (set_stream pairs: torch.cuda.stream()'s constructor and __enter__ each
    resolve the device through hipGetDeviceCount, ~0.1 ms apiece); a no-op for ``s is None``."""

    def __init__(self, s_):
        self.s = s_

    def __enter__(self):
        if self.s is not None:
            self.prev = torch.cuda.current_stream(self.s.device)
            torch.cuda.set_stream(self.s)
        return self

    def __exit__(self, *exc):
        if self.s is not None:
            torch.cuda.set_stream(self.prev)
        return False


PROFILE = __import__("os").environ.get("TRS_DIST_PROFILE", "0") == "1"
phase_events = {}       # phase -> [(start event, end event)]   (TRS_DIST_PROFILE=1; read by phase_times_ms())
wire_bytes = {"ids": 0, "rows_fwd": 0, "rows_bwd": 0, "steps": 0}      # bytes this rank SENT to other ranks


def _pn(name: str, weight: torch.Tensor) -> str:
    """phase name with the table's width: the E = 64 table and its E = 1 first-order companion are timed apart"""
    return f"{name} [E={weight.shape[1]}]" if PROFILE else name


class _phase:
    """``with _phase("row a2a"):`` brackets the enqueued work with a HIP event pair on the current stream when profiling"""

    def __init__(self, name, dev):
        self.on = PROFILE and dev.type == "cuda"
        self.name = name

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            phase_events.setdefault(self.name, []).append((self.a, b))
        return False


def phase_times_ms(reset: bool = True):
    """mean device milliseconds per recorded phase (synchronises)"""
    if not phase_events:
        return {}
    torch.cuda.synchronize()
    out = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in phase_events.items()}
    if reset:
        phase_events.clear()
    return out


def _count_wire(kind: str, splits, rank: int, row_bytes: int, total_rows: int, world: int):
    if world == 1:
        return
    if splits is None:                   # equal splits: everything but this rank's own slots travels
        wire_bytes[kind] += (total_rows - total_rows // world) * row_bytes
    else:
        wire_bytes[kind] += (sum(splits) - splits[rank]) * row_bytes


def pad_slots(counts: torch.Tensor, send_ids: torch.Tensor, send_pos: torch.Tensor, inv_pos: torch.Tensor, cap: int,
              world: int):
    """Compact per-owner lists -> ``world`` slots of ``cap`` entries each, on the device, with no host read.
    Returns (send_ids_pad (W*cap) int32, PAD-filled; pos_pad (W*cap) int32: the lookup that sits in each slot or PAD;
    inv_pad (B*N) int32: the slot of every lookup -- W*cap, one past the last slot, for a lookup that did not fit;
    bad (0-d bool): some owner's share exceeded ``cap`` or a row id was negative).  Index plumbing only: cumsum /
    bucketize / scatter on B*N int32 values."""
    K = send_ids.numel()
    dev = send_ids.device
    ends = counts.cumsum(0)
    starts = ends - counts
    k = torch.arange(K, device=dev)
    owner = torch.bucketize(k, ends, right=True).clamp_max(world - 1)
    within = k - starts[owner]
    total = world * cap
    slot = torch.where(within < cap, owner * cap + within, torch.full_like(k, total))     # `total` = the trash slot
    ids_pad = torch.full((total + 1,), PAD, dtype=torch.int32, device=dev).scatter_(0, slot, send_ids)
    pos_pad = torch.full((total + 1,), PAD, dtype=torch.int32, device=dev).scatter_(0, slot, send_pos)
    inv_pad = slot.to(torch.int32)[inv_pos.long()]
    bad = (counts.max() > cap) | (send_ids.min() < 0) if K else torch.zeros((), dtype=torch.bool, device=dev)
    return ids_pad[:total], pos_pad[:total], inv_pad, bad


class RoutePlan:
    """Everything about one batch of indices that does not depend on the table: who owns each lookup, the
    per-peer split sizes (host ints) and the local row ids every peer asked this rank for.  Tables looked up
    with the same index tensor (the E=64 embeddings and the E=1 first-order weights of one model) share it,
    so the bucketing, the count exchange (one host sync) and the id all-to-all happen once per batch."""
    __slots__ = ("send_pos", "inv_pos", "send_splits", "recv_splits", "recv_ids", "cap", "event", "stream",
                 "send_ids", "self_lo", "self_n", "recv_lo", "_owner_ids")

    def owner_ids(self, local_direct: bool) -> torch.Tensor:
        """the local row ids the owner-side reduction runs over, in the order of the gradient rows it receives: with
        ``local_direct`` the rows that arrive over the wire (``recv_ids`` without this rank's own segment) followed by the
        lookups this rank owns itself, whose gradient rows never travel; ``recv_ids`` otherwise"""
        if not local_direct:
            return self.recv_ids
        if self._owner_ids is None:
            lo, n, rlo = self.self_lo, self.self_n, self.recv_lo
            if self.recv_ids.numel() == n:          # one rank: everything is local
                self._owner_ids = self.send_ids[lo:lo + n]
            else:
                self._owner_ids = torch.cat([self.recv_ids[:rlo], self.recv_ids[rlo + n:], self.send_ids[lo:lo + n]])
        return self._owner_ids

    def use_on(self, cur) -> None:
        """A plan built on the communication stream (prefetch_lookup) sits in the shared route cache: a later hit from
        another stream -- a second table looked up with the same indices, or an inline forward after the prefetched rows
        were evicted -- must wait for the producing stream and tell the allocator about its use of the plan's tensors."""
        ev = getattr(self, "event", None)
        if ev is None or cur is None or cur == self.stream:
            return
        cur.wait_event(ev)
        for t in (self.send_pos, self.inv_pos, self.recv_ids, self.send_ids, self._owner_ids):
            if t is not None and t.is_cuda:
                t.record_stream(cur)


_route_cache: List[tuple] = []      # [(key, idx kept alive, RoutePlan)]


class _PendingRoute:
    """First half of a route plan (owner bucketing + count exchange), started early for the NEXT batch -- the way
    a data loader prefetches -- so that the host-side read of the split sizes finds them already copied to pinned
    memory and never stalls the launch queue behind the current step's work."""
    __slots__ = ("send_ids", "send_pos", "inv_pos", "host_counts", "ready", "side_event", "side_stream")


def _route_key(idx, mod):
    return (idx.data_ptr(), idx._version, tuple(idx.shape), idx.dtype, mod.route_key)


_pending_routes: List[tuple] = []    # [(key, idx kept alive, _PendingRoute)]
MAX_PENDING_ROUTES = 4               # batches routed ahead of their forward pass
route_stats = {"prefetched": 0, "cached": 0, "cold": 0}     # how forward passes obtained their route plan


def clear_route_caches():
    _route_cache.clear()
    _pending_routes.clear()


F_._clear_hooks.append(clear_route_caches)      # F_.clear_caches() (GraphedStep) drops the route plans too
F_._clear_hooks.append(lambda: setattr(HipOps, "_uniq_cache", None))


def _start_route(idx: torch.Tensor, mod) -> "_PendingRoute":
    ops, group, world = mod.ops, mod.group, mod.world
    pr = _PendingRoute()
    pr.side_event = pr.side_stream = None
    if mod.dedup:
        counts, send_ids, inv = ops.unique_route(idx, mod.offsets, mod.rows_per_rank, world)
        pr.send_ids, pr.send_pos, pr.inv_pos = send_ids, None, inv
    else:
        counts, send_ids, send_pos, inv_pos = ops.bucket_by_owner(idx, mod.offsets, mod.rows_per_rank, world)
        pr.send_ids, pr.send_pos, pr.inv_pos = send_ids, send_pos, inv_pos
    if world == 1:
        pr.host_counts, pr.ready = None, None        # one rank: everything stays local, nothing to read back
        return pr
    if mod.capacity is not None and not mod.dedup:
        cap = mod.slot_capacity(idx.numel())
        pr.send_ids, pr.send_pos, pr.inv_pos, bad = pad_slots(counts, send_ids, send_pos, inv_pos, cap, world)
        flag = F_._ErrFlag(idx.device)
        flag.t.copy_(torch.maximum(flag.t, bad.to(torch.int32).reshape(1)))
        flag.check("sharded lookup (fixed-capacity slots)")
        pr.host_counts, pr.ready = None, None        # equal splits: no count exchange, no host read
        return pr
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts, group=group)
    both = torch.stack([counts, recv_counts])
    if both.is_cuda:
        pr.host_counts = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
        pr.host_counts.copy_(both, non_blocking=True)
        pr.ready = torch.cuda.Event()
        pr.ready.record()
    else:
        pr.host_counts, pr.ready = both, None
    return pr


ROUTE_SIDE = __import__("os").environ.get("TRS_ROUTE_SIDE", "1") != "0"


def prefetch_route(idx: torch.Tensor, mod) -> None:
    """Start routing ``idx`` (a batch that will be looked up soon) now; the next ``forward`` picks it up.  On a HIP
    device the owner bucketing runs on the "route" side stream, beside whatever the caller's stream is busy with (it
    depends on the indices only); the forward that picks the plan up waits for its event."""
    idx = idx.rename(None) if idx.has_names() else idx
    if idx.dtype not in (torch.int64, torch.int32):
        idx = idx.long()
    idx = idx.contiguous()
    key = _route_key(idx, mod)
    if any(k == key for k, _, _ in _pending_routes) or any(k == key for k, _, _ in _route_cache):
        return
    if idx.is_cuda and ROUTE_SIDE and (mod.world == 1 or (mod.capacity is not None and not mod.dedup)):
        # (the exact-split mode issues a count exchange and a host copy from inside _start_route: it keeps the caller's
        # stream, where it is ordered with the other collectives)
        pr, ev, side = F_.run_on_side(idx.device, "route", lambda: _start_route(idx, mod))
        idx.record_stream(side)
        pr.side_event, pr.side_stream = ev, side
    else:
        pr = _start_route(idx, mod)
    _pending_routes.append((key, idx, pr))
    if len(_pending_routes) > MAX_PENDING_ROUTES:
        _pending_routes.pop(0)


def _route_plan(idx: torch.Tensor, mod) -> "RoutePlan":
    key = _route_key(idx, mod)
    for k, _, p in _route_cache:
        if k == key:
            route_stats["cached"] += 1
            p.use_on(torch.cuda.current_stream(idx.device) if idx.is_cuda else None)
            return p
    pr = None
    for i, (k, _, cand) in enumerate(_pending_routes):
        if k == key:
            pr = cand
            _pending_routes.pop(i)
            route_stats["prefetched"] += 1
            break
    if pr is None:
        route_stats["cold"] += 1
        with _phase("route (bucket by owner)", idx.device):
            pr = _start_route(idx, mod)
    elif pr.side_event is not None:
        cur = torch.cuda.current_stream(idx.device)
        if cur != pr.side_stream:          # routed ahead on the side stream: its tensors are used on this one from here on
            cur.wait_event(pr.side_event)
            for t in (pr.send_ids, pr.send_pos, pr.inv_pos):
                if t is not None:
                    t.record_stream(cur)
    p = RoutePlan()
    p.send_pos, p.inv_pos, p.cap = pr.send_pos, pr.inv_pos, 0
    p.send_ids, p._owner_ids = pr.send_ids, None
    p.event, p.stream = None, (torch.cuda.current_stream(idx.device) if idx.is_cuda else None)
    if pr.host_counts is None and mod.world > 1:
        # fixed-capacity slots: equal splits, nothing to read on the host
        p.cap = mod.slot_capacity(idx.numel())
        p.send_splits = p.recv_splits = None
        p.recv_ids = torch.empty_like(pr.send_ids)
        with _phase("id all-to-all", idx.device):
            dist.all_to_all_single(p.recv_ids, pr.send_ids, group=mod.group)
        _count_wire("ids", None, mod.rank, 4, pr.send_ids.numel(), mod.world)
    elif mod.world == 1:
        n = int(pr.send_ids.numel())                 # a shape, not a device value: no synchronisation
        p.send_splits, p.recv_splits, p.recv_ids = [n], [n], pr.send_ids
        if FORCE_COLLECTIVES and pr.send_ids.is_cuda:
            p.recv_ids = torch.empty_like(pr.send_ids)
            dist.all_to_all_single(p.recv_ids, pr.send_ids, group=mod.group)
    else:
        if pr.ready is not None:
            pr.ready.synchronize()                 # waits for the tiny count copy only (long done when prefetched)
        p.send_splits = pr.host_counts[0].tolist()
        p.recv_splits = pr.host_counts[1].tolist()
        p.recv_ids = torch.empty(sum(p.recv_splits), dtype=torch.int32, device=idx.device)
        with _phase("id all-to-all", idx.device):
            _all_to_all(p.recv_ids, pr.send_ids, p.recv_splits, p.send_splits, mod.group)
        _count_wire("ids", p.send_splits, mod.rank, 4, 0, mod.world)
    # this rank's own segment of the exchange order (send side) and where it sits among the ids it receives
    if mod.world == 1:
        p.self_lo, p.self_n, p.recv_lo = 0, int(pr.send_ids.numel()), 0
    elif p.cap:
        p.self_lo, p.self_n, p.recv_lo = mod.rank * p.cap, p.cap, mod.rank * p.cap
    else:
        p.self_lo, p.self_n = sum(p.send_splits[:mod.rank]), p.send_splits[mod.rank]
        p.recv_lo = sum(p.recv_splits[:mod.rank])
    _route_cache.append((key, idx, p))
    if len(_route_cache) > 2:
        _route_cache.pop(0)
    return p


def _self_zero(splits, rank: int, world: int, cap: int):
    """split sizes of an exchange whose self segment does not travel: the given per-peer sizes (or ``cap`` each) with
    this rank's own entry zeroed"""
    out = list(splits) if splits is not None else [cap] * world
    out[rank] = 0
    return out


def _exchange(out_rows: int, inp: torch.Tensor, out_splits, in_splits, mod, zero_row: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """all-to-all of row blocks; a one-rank group hands the tensor through.  ``zero_row`` (explicit splits): one more,
    zeroed, row behind the received ones; ``out``: receive into this (out_rows, E) tensor."""
    if mod.world == 1 and not (FORCE_COLLECTIVES and inp.is_cuda):
        return inp
    if out_splits is not None and (zero_row or out is not None):
        if out is None:
            buf = torch.empty(out_rows + 1, inp.shape[1], dtype=inp.dtype, device=inp.device)
            buf[out_rows:].zero_()
            out = buf[:out_rows]
        else:
            buf = out
        _all_to_all(out, inp, out_splits, in_splits, mod.group)
        return buf
    if out_splits is None:            # fixed-capacity slots: one more (zero) row behind the received ones for lookups
        buf = torch.empty(out_rows + 1, inp.shape[1], dtype=inp.dtype, device=inp.device)      # that did not fit
        buf[out_rows:].zero_()
        dist.all_to_all_single(buf[:out_rows], inp, group=mod.group)
        return buf
    out = torch.empty(out_rows, inp.shape[1], dtype=inp.dtype, device=inp.device)
    _all_to_all(out, inp, out_splits, in_splits, mod.group)
    return out


def _fetch_rows(weight: torch.Tensor, idx: torch.Tensor, mod, local: Optional[bool] = None):
    """Steps 1-4 of the forward: route plan, owner-side gather, rows back.  -> (plan, rows in exchange order; with
    ``local`` (default: the module's ``local_direct``) without this rank's own segment)"""
    ops = mod.ops
    local = mod.local_direct if local is None else local
    plan = _route_plan(idx, mod)
    padded = plan.cap > 0
    n_valid = mod.row_range[1] - mod.row_range[0]
    row_bytes = weight.shape[1] * weight.element_size()
    if local:
        # the lookups this rank owns itself never travel: gathered for the OTHER requesters only, exchanged with a zero
        # self split; forward reads its own rows straight from the shard (ops.unpermute_local)
        n, rlo = plan.self_n, plan.recv_lo
        Kr = plan.recv_ids.numel() - n
        rows = torch.empty(Kr, weight.shape[1], dtype=weight.dtype, device=weight.device)
        with _phase(_pn("owner gather", weight), idx.device):
            if rlo:
                ops.gather_local(weight, plan.recv_ids[:rlo], n_valid, padded=padded, out=rows[:rlo])
            if Kr - rlo:
                ops.gather_local(weight, plan.recv_ids[rlo + n:], n_valid, padded=padded, out=rows[rlo:])
        with _phase(_pn("row all-to-all (forward)", weight), idx.device):
            if mod.world == 1:
                back = rows
            else:
                outs = _self_zero(plan.send_splits, mod.rank, mod.world, plan.cap)
                ins = _self_zero(plan.recv_splits, mod.rank, mod.world, plan.cap)
                back = _exchange(sum(outs), rows, outs, ins, mod, zero_row=padded)
                _count_wire("rows_fwd", ins, mod.rank, row_bytes, 0, mod.world)
        return plan, back
    with _phase(_pn("owner gather", weight), idx.device):
        rows = ops.gather_local(weight, plan.recv_ids, n_valid, padded=True) if padded else \
            ops.gather_local(weight, plan.recv_ids, n_valid)                                    # rows of my shard
    with _phase(_pn("row all-to-all (forward)", weight), idx.device):
        if padded:
            back = _exchange(plan.recv_ids.numel(), rows, None, None, mod)
        else:
            back = _exchange(sum(plan.send_splits), rows, plan.send_splits, plan.recv_splits, mod)
    _count_wire("rows_fwd", None if padded else plan.recv_splits, mod.rank, row_bytes, plan.recv_ids.numel(), mod.world)
    return plan, back


class _PrefetchedLookup:
    """rows of a batch fetched ahead of its forward (prefetch_lookup): the plan, the received rows, the event on the
    communication stream after which they are complete"""
    __slots__ = ("plan", "back", "event", "version", "stale_ok", "local")


_lookup_cache: List[tuple] = []      # [(key, idx kept alive, weight id, _PrefetchedLookup)]
MAX_PREFETCHED_LOOKUPS = 4
lookup_stats = {"prefetched": 0, "inline": 0, "stale_dropped": 0}
F_._clear_hooks.append(_lookup_cache.clear)


def prefetch_lookup(idx: torch.Tensor, mod, stale_ok: bool = False) -> None:
    """Run the exchange half of the lookup of ``idx`` (a batch whose forward comes NEXT) now, on the communication stream,
    so that it overlaps whatever the compute stream is busy with; the matching ``forward`` waits for its event and only
    un-permutes the rows.  See the module docstring for the staleness rule with a fused optimizer."""
    if mod.fused_optimizer is not None and not stale_ok:
        raise RuntimeError("prefetch_lookup: with a fused optimizer on the owner an early lookup reads rows that are one "
                           "update behind; pass stale_ok=True to accept that, or look up at forward time")
    idx = idx.rename(None) if idx.has_names() else idx
    if idx.dtype not in (torch.int64, torch.int32):
        idx = idx.long()
    idx = idx.contiguous()
    key = _route_key(idx, mod)
    weight = mod.embedding.weight
    if any(k == key and w == id(weight) for k, _, w, _ in _lookup_cache):
        return
    cs = comm_stream(idx.device)
    if cs is not None:
        cs.wait_stream(torch.cuda.current_stream(idx.device))     # the indices (and any update of the shard) come first
    with _on_stream(cs), torch.no_grad():
        pl = _PrefetchedLookup()
        # an early lookup under a fused owner-side optimizer (stale_ok) must read EVERY row as of now -- one update behind,
        # consistently: the rows this rank owns itself go through the buffers too instead of being read at forward time
        pl.local = mod.local_direct and mod.fused_optimizer is None
        pl.plan, pl.back = _fetch_rows(weight.detach(), idx, mod, pl.local)
        pl.event = None
        if cs is not None:
            pl.event = torch.cuda.Event()
            pl.event.record(cs)
            if pl.plan.stream == cs and pl.plan.event is None:
                pl.plan.event = pl.event      # the plan was built here: other streams that hit it in the route cache wait
    # what the rows were read from: an in-place update of the shard between this prefetch and the forward (optimizer.step()
    # bumps the version; the fused owner-side optimizer writes through .data and is covered by ``stale_ok`` above) makes the
    # forward drop these rows and look up again, unless staleness was accepted
    pl.version, pl.stale_ok = weight._version, bool(stale_ok)
    _lookup_cache.append((key, idx, id(weight), pl))
    if len(_lookup_cache) > MAX_PREFETCHED_LOOKUPS:
        _lookup_cache.pop(0)


def _take_prefetched(idx: torch.Tensor, weight: torch.Tensor, mod):
    key = _route_key(idx, mod)
    for i, (k, _, w, pl) in enumerate(_lookup_cache):
        if k == key and w == id(weight):
            _lookup_cache.pop(i)
            if pl.version != weight._version and not pl.stale_ok:
                lookup_stats["stale_dropped"] += 1
                return None
            return pl
    return None


class _ShardedLookup(Function):
    """(local shard, local batch of indices) -> (B,N,E) block [+ FM]; gradient flows back to the shard owners."""

    @staticmethod
    def forward(ctx, weight, idx, mod):
        ops = mod.ops
        B, N = idx.shape
        pl = _take_prefetched(idx, weight, mod)
        if pl is not None:
            lookup_stats["prefetched"] += 1
            plan, back, local = pl.plan, pl.back, pl.local
            if pl.event is not None:
                cur = torch.cuda.current_stream(idx.device)
                cur.wait_event(pl.event)
                for t in (back, plan.inv_pos, plan.send_pos, plan.recv_ids, plan.send_ids):      # made on the communication stream, used here
                    if t is not None:
                        t.record_stream(cur)
        else:
            lookup_stats["inline"] += 1
            mod.order_after_grad()      # an owner-side update / reduction still running on the communication stream
            local = mod.local_direct
            plan, back = _fetch_rows(weight, idx, mod, local)
        padded = plan.cap > 0
        with _phase(_pn("un-permute (+FM)", weight), idx.device):
            bufs = {"out": mod.output_buffers(B, N, weight)} if mod.persistent_outputs else {}
            if local:
                block, fm, fm_sum = ops.unpermute_local(back, weight.detach(), mod.row_range[1] - mod.row_range[0],
                                                        plan.inv_pos, plan.send_ids, plan.self_lo, plan.self_n, B, N,
                                                        mod.fuse_fm, **bufs)
            else:
                block, fm, fm_sum = ops.unpermute(back, plan.inv_pos, B, N, mod.fuse_fm, **bufs)
        # own_direct: dense shard gradient, no fused optimizer -- this rank's own lookups are reduced straight from the
        # block gradient by the unsharded backward (ops.own_grad_dense); only rows that arrived over the wire go through
        # the permuted / exchanged path
        own_direct = (local and mod.fused_optimizer is None and weight.shape[0] <= mod.dense_grad_max_rows
                      and hasattr(ops, "own_grad_dense") and mod.own_direct)
        owner_ids = plan.owner_ids(local)
        if own_direct:
            remote_ids = owner_ids[: owner_ids.numel() - plan.self_n]
            ops.prefetch_own_buckets(weight, idx, mod.offsets_local)
            if remote_ids.numel() and hasattr(ops, "prefetch_owner_buckets"):
                ops.prefetch_owner_buckets(weight, remote_ids, padded, pipelined=bool(mod.overlap_grad_exchange))
        elif weight.shape[0] <= mod.dense_grad_max_rows and hasattr(ops, "prefetch_owner_buckets"):
            ops.prefetch_owner_buckets(weight, owner_ids, padded, pipelined=bool(mod.overlap_grad_exchange))
        ctx.mod = mod
        ctx.padded = padded
        ctx.splits = (plan.send_splits, plan.recv_splits)
        ctx.local = (plan.self_lo, plan.self_n, plan.cap) if local else None
        ctx.own_direct = own_direct
        ctx.save_for_backward(weight, owner_ids, plan.send_pos if plan.send_pos is not None else plan.inv_pos,
                              block if mod.fuse_fm else None, fm_sum, back if mod.dedup else None,
                              idx if own_direct else None)
        ctx.set_materialize_grads(False)
        if fm is None:
            fm = torch.empty(0, dtype=block.dtype, device=block.device)
            ctx.mark_non_differentiable(fm)
        return block, fm

    @staticmethod
    @once_differentiable
    def backward(ctx, g_block, g_fm):
        mod = ctx.mod
        ops = mod.ops
        weight, recv_ids, pos, block, fm_sum, back, idx = ctx.saved_tensors
        send_splits, recv_splits = ctx.splits
        if g_block is None and g_fm is None:
            return None, None, None
        dev = weight.device
        if ctx.local is not None:
            return _ShardedLookup._backward_local(ctx, g_block, g_fm, weight, recv_ids, pos, block, fm_sum, idx)
        with _phase(_pn("permute gradient", weight), dev):
            if mod.dedup:
                # one gradient row per DISTINCT row of the local batch (duplicates summed before they travel)
                g_rows = ops.reduce_grad_unique(g_block, pos, back, g_fm if mod.fuse_fm else None, fm_sum)
            else:
                if block is None:
                    block = g_block     # shape carrier only
                g_rows = ops.permute_grad(g_block, pos, g_fm if mod.fuse_fm else None, fm_sum, block)
        # The reverse exchange and the owner-side reduction run on the communication stream when asked to overlap with
        # the next batch's forward; never when a gradient is being ACCUMULATED (autograd would add into .grad on the
        # compute stream right away).
        cs = comm_stream(dev) if (mod.overlap_grad_exchange and weight.grad is None) else None
        if cs is not None:
            ev = torch.cuda.Event()
            ev.record()
            cs.wait_event(ev)
            for t in (g_rows, recv_ids):
                t.record_stream(cs)
        row_bytes = g_rows.shape[1] * g_rows.element_size()
        with _on_stream(cs):
            with _phase(_pn("row all-to-all (gradient)", weight), dev):
                if ctx.padded:  # equal splits; padding slots carry zero rows (pos = PAD) and update nothing on the owner
                    recv_g = _exchange(g_rows.shape[0], g_rows, None, None, mod)[: g_rows.shape[0]]
                else:
                    recv_g = _exchange(sum(recv_splits), g_rows, recv_splits, send_splits, mod)      # reverse exchange
            _count_wire("rows_bwd", None if ctx.padded else send_splits, mod.rank, row_bytes, g_rows.shape[0], mod.world)
            wire_bytes["steps"] += 1
            with _phase(_pn("owner reduce / update", weight), dev):
                gw = _ShardedLookup._owner_reduce(mod, weight, recv_ids, recv_g, ctx.padded)
            if cs is not None:
                mod._grad_event = torch.cuda.Event()
                mod._grad_event.record(cs)
        return gw, None, None

    @staticmethod
    def _owner_reduce(mod, weight, ids, rows, padded):
        """the owner-side end of the gradient path: fused optimizer step, dense shard gradient or sparse COO gradient"""
        ops = mod.ops
        dense_index = weight.shape[0] <= mod.dense_grad_max_rows
        pad_kw = {"padded": True} if padded else {}
        if mod.fused_optimizer is not None:
            # the owner steps its rows right here: no gradient tensor of any kind (weight.grad stays None); small shards
            # through a bucket index over all their rows, large ones through the compact list of distinct touched rows
            # (torch.unique, a sort of the B*N ids)
            ops.shard_update(weight, ids, rows, mod.fused_optimizer, weight.shape[0] <= mod.dense_index_max_rows, **pad_kw)
            return None
        if dense_index:
            return ops.shard_grad_dense(weight, ids, rows, **pad_kw)
        ids = ids.clamp_min(0) if padded else ids          # padding: zero rows added to row 0
        return torch.sparse_coo_tensor(ids.long().unsqueeze(0), rows, size=weight.shape)

    @staticmethod
    def _backward_local(ctx, g_block, g_fm, weight, owner_ids, pos, block, fm_sum, idx=None):
        """backward with ``local_direct``: only the gradient rows of lookups OTHER ranks own are permuted into exchange
        order and sent; the rows of this rank's own lookups are written straight behind the received ones (G = [rows that
        arrived | own rows], the order of ``RoutePlan.owner_ids``) and the owner-side reduction runs over G."""
        mod = ctx.mod
        ops = mod.ops
        dev = weight.device
        lo, n, cap = ctx.local
        send_splits, recv_splits = ctx.splits
        if block is None:
            block = g_block         # shape carrier only
        gf = g_fm if mod.fuse_fm else None
        E = block.shape[-1]
        Ks = pos.numel()
        Kr = owner_ids.numel() - n
        own = ctx.own_direct
        G = torch.empty(Kr + (0 if own else n), E, dtype=block.dtype, device=dev)
        g_send = None
        with _phase(_pn("permute gradient", weight), dev):
            if Ks - n:
                g_send = torch.empty(Ks - n, E, dtype=block.dtype, device=dev)
                if lo:
                    ops.permute_grad(g_block, pos[:lo], gf, fm_sum, block, out=g_send[:lo])
                if Ks - n - lo:
                    ops.permute_grad(g_block, pos[lo + n:], gf, fm_sum, block, out=g_send[lo:])
            if n and not own:
                ops.permute_grad(g_block, pos[lo:lo + n], gf, fm_sum, block, out=G[Kr:])
        gw_own = None
        if own:
            with _phase(_pn("own lookups: bucket walk", weight), dev):
                gw_own = ops.own_grad_dense(weight, idx, mod.offsets_local, g_block, gf, fm_sum if gf is not None else None)
        cs = comm_stream(dev) if (mod.overlap_grad_exchange and weight.grad is None) else None
        if cs is not None:
            ev = torch.cuda.Event()
            ev.record()
            cs.wait_event(ev)
            for t in (G, g_send, owner_ids, gw_own):
                if t is not None:
                    t.record_stream(cs)
        with _on_stream(cs):
            if mod.world > 1:
                with _phase(_pn("row all-to-all (gradient)", weight), dev):
                    ins = _self_zero(send_splits, mod.rank, mod.world, cap)       # what this rank sends back to each owner
                    outs = _self_zero(recv_splits, mod.rank, mod.world, cap)      # what it receives as an owner
                    _exchange(Kr, g_send, outs, ins, mod, out=G[:Kr])
                _count_wire("rows_bwd", ins, mod.rank, E * G.element_size(), 0, mod.world)
            wire_bytes["steps"] += 1
            with _phase(_pn("owner reduce / update", weight), dev):
                if own:
                    gw = gw_own
                    ops.accumulate_rows(gw, owner_ids[:Kr], G, ctx.padded)
                else:
                    gw = _ShardedLookup._owner_reduce(mod, weight, owner_ids, G, ctx.padded)
            if cs is not None:
                mod._grad_event = torch.cuda.Event()
                mod._grad_event.record(cs)
        return gw, None, None


class RowShardedMultiIndicesEmbedding(BaseInput):
    """``MultiIndicesEmbedding`` (multi_indices_emb.py:18-112) with the table row-sharded over the process
    group: same constructor core (embed_size, field_sizes, flatten), same forward contract ((B,N) local
    indices -> (B,N,E) named ('B','N','E')).  ``embedding.weight`` holds THIS rank's rows
    [rank*rows_per_rank, ...); ``full_state_dict()`` / ``load_full_weight()`` convert to/from the unsharded
    ``embedding.weight`` of the reference."""

    def __init__(self, embed_size: int, field_sizes: List[int], flatten: bool = False, fuse_fm: bool = False,
                 dtype: torch.dtype = torch.float32, device='cpu', process_group=None, ops=None,
                 dense_grad_max_rows: int = DENSE_GRAD_MAX_ROWS, dedup: bool = False,
                 capacity: Optional[float] = None, overlap_grad_exchange: bool = False,
                 persistent_outputs: bool = False, local_direct: Optional[bool] = None):
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("RowShardedMultiIndicesEmbedding needs torch.distributed to be initialised")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.ops = ops if ops is not None else HipOps()
        self.num_rows = int(sum(field_sizes))
        self.rows_per_rank, ranges = shard_ranges(self.num_rows, self.world)
        self.row_range = ranges[self.rank]
        n_local = max(self.row_range[1] - self.row_range[0], 1)
        self.embedding = nn.Embedding(n_local, embed_size, device=device, dtype=dtype)
        self.register_buffer('offsets', field_offsets(field_sizes).to(device), persistent=False)
        self.flatten = flatten
        self.fuse_fm = fuse_fm and not flatten
        self.field_size = self.num_rows
        self.embed_size = embed_size
        self.padding_idx = None
        self.dense_grad_max_rows = dense_grad_max_rows
        self.dedup = bool(dedup)
        if capacity is not None and capacity < 1.0:
            raise ValueError("capacity is a factor on the even share B*N/world: it must be >= 1")
        if capacity is not None and dedup:
            raise ValueError("fixed-capacity slots and dedup (a data-dependent number of rows) exclude each other")
        self.capacity = None if capacity is None else float(capacity)
        self.overlap_grad_exchange = bool(overlap_grad_exchange)
        # persistent_outputs: every forward writes its (B,N,E) block (and FM term) into the SAME buffers -- what lets the
        # dense part of the step behind the lookup be replayed from a hipGraph (graph.GraphedRegion reads fixed addresses)
        # while the exchanges stay eager on their own stream.  The caller must be done with step k's outputs before step
        # k+1's forward (true for a training loop: backward(k) precedes forward(k+1) on the compute stream).
        self.persistent_outputs = bool(persistent_outputs)
        self._out_bufs = {}
        self._grad_event = None
        self._caps = {}
        # local_direct: lookups this rank owns itself are read straight from its shard by the un-permute kernel and their
        # gradient rows are written straight into the owner-side reduction's input -- no gather into a send buffer, no
        # self copy inside the all-to-alls (TRS_SHARD_LOCAL_DIRECT=0: everything through the exchange buffers, the round-5
        # arrangement; dedup keeps it: its backward needs the received rows of every distinct id)
        self.local_direct = bool(local_direct if local_direct is not None else
                                 __import__("os").environ.get("TRS_SHARD_LOCAL_DIRECT", "1") != "0") \
            and hasattr(self.ops, "unpermute_local") and not self.dedup
        # own_direct: with a dense shard gradient and no fused optimizer, this rank's own lookups are reduced by the unsharded
        # backward straight from the block gradient (TRS_SHARD_OWN_DIRECT=0: permuted and reduced with the received rows)
        self.own_direct = __import__("os").environ.get("TRS_SHARD_OWN_DIRECT", "1") != "0"
        # fused optimizer: shards up to this many rows are stepped through a bucket index over ALL their rows, larger ones
        # through the compact list of distinct touched rows (_owner_reduce).  TRS_SHARD_DENSE_INDEX_ROWS raises the limit;
        # measured at 125 M rows (round 6, one rank, Adagrad): index over all rows 6.37 ms per step (owner update 2.85 ms:
        # zero + scan + walk over 125 M mostly empty rows), compact rows 3.04 ms (0.89 ms) -- the default stays
        self.dense_index_max_rows = max(int(dense_grad_max_rows), int(__import__("os").environ.get(
            "TRS_SHARD_DENSE_INDEX_ROWS", "0"))) if dense_grad_max_rows > 0 else 0
        self.register_buffer('offsets_local', self.offsets - self.row_range[0], persistent=False)
        self.route_key = (tuple(int(f) for f in field_sizes), self.world, id(process_group), self.dedup, self.capacity)
        self.length = embed_size * len(field_sizes) if flatten else embed_size

    def output_buffers(self, B: int, N: int, weight: torch.Tensor):
        key = (B, N, weight.dtype)
        bufs = self._out_bufs.get(key)
        if bufs is None:
            E, dev = weight.shape[1], weight.device
            block = torch.empty(B, N, E, dtype=weight.dtype, device=dev)
            fm = torch.empty(B, E, dtype=weight.dtype, device=dev) if self.fuse_fm else None
            fm_sum = torch.empty(B, E, dtype=torch.float32, device=dev) if self.fuse_fm else None
            if len(self._out_bufs) >= 4:
                self._out_bufs.clear()
            bufs = self._out_bufs[key] = (block, fm, fm_sum)
        # fresh aliases every call: the caller's tensor objects receive names / a grad_fn (forward sets ``out.names``), the
        # buffers themselves must stay plain
        return tuple(None if t is None else t.view(t.shape) for t in bufs)

    def slot_capacity(self, lookups: int) -> int:
        """slots per peer of a fixed-capacity exchange of ``lookups`` = B*N row ids (a multiple of 64).  Equal splits
        need the same slot size on every rank, i.e. the same local batch size: checked ONCE per distinct size (one
        small all-reduce the first time a size is seen), never per step."""
        cap = self._caps.get(lookups)
        if cap is None:
            t = torch.tensor([lookups, -lookups], dtype=torch.int64, device=self.embedding.weight.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            hi, lo = int(t[0]), -int(t[1])
            if hi != lo:
                raise RuntimeError(f"fixed-capacity exchange: every rank must look up the same number of ids per step "
                                   f"(this step: between {lo} and {hi})")
            even = -(-lookups // self.world)
            cap = self._caps[lookups] = max(64, -(-int(even * self.capacity) // 64) * 64)
        return cap

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        idx = inputs.rename(None) if inputs.has_names() else inputs
        if idx.dim() != 2 or idx.shape[1] != self.offsets.numel():
            raise ValueError(f'inputs must be (B, {self.offsets.numel()}), got {tuple(idx.shape)}')
        if idx.dtype not in (torch.int64, torch.int32):
            idx = idx.long()
        out, fm = _ShardedLookup.apply(self.embedding.weight, idx.contiguous(), self)
        if self.fuse_fm:
            out._trs_fused_fm = (fm, out._version)
        if self.flatten:
            out = out.reshape(out.shape[0], 1, -1)
        out.names = ('B', 'N', 'E',)
        return out

    def prefetch_route(self, next_inputs: torch.Tensor) -> None:
        """Hint: ``next_inputs`` is an index batch that will be looked up soon.  Owner bucketing and the count exchange
        start now, so that forward does not block the host on the split sizes (see _PendingRoute).  Hinting TWO steps
        ahead keeps the host from ever waiting on the device: with a one-step hint the count copy sits in the queue
        behind the current forward, and a host that waits for it falls into lock-step with the GPU (measured on the
        world-size-1 run: 2.9 ms/step when the host stays ahead, 4.6 ms when it does not -- bistable)."""
        prefetch_route(next_inputs, self)

    def prefetch_lookup(self, next_inputs: torch.Tensor, stale_ok: bool = False) -> None:
        """Hint: ``next_inputs`` is the index batch of the NEXT forward.  Its route, id exchange, owner-side gather and
        row exchange start now on the communication stream (``dist.prefetch_lookup``) and overlap the current batch's
        dense compute.  Bit-identical to the lookup at forward time as long as the shard is not updated in between: an
        ``optimizer.step()`` on the shard between this call and the forward makes the forward drop the early rows and look
        up again (``lookup_stats['stale_dropped']``) -- hint AFTER the step, or pass ``stale_ok=True`` to keep rows that
        are one update behind."""
        prefetch_lookup(next_inputs, self, stale_ok)

    def wait_grad(self) -> None:
        """With ``overlap_grad_exchange``: make the current stream wait for the last backward's gradient exchange and
        owner-side reduction (``embedding.weight.grad`` / the fused update).  Call before reading ``.grad`` (an optimizer
        step, clipping, a checkpoint).  Lookups order themselves: a prefetched one runs on the communication stream behind
        the update, an inline one (forward without a matching prefetch) waits for the same event -- ``order_after_grad``."""
        ev, self._grad_event = self._grad_event, None
        if ev is not None:
            torch.cuda.current_stream(self.embedding.weight.device).wait_event(ev)

    def order_after_grad(self) -> None:
        """Every reader / writer of the shard on the CURRENT stream (an inline lookup, load_full_weight, full_weight,
        state_dict) comes behind the last overlapped gradient exchange: with a fused optimizer on the owner that exchange
        ends in an in-place update of the shard on the communication stream.  Does not consume the event (wait_grad does)."""
        ev = self._grad_event
        if ev is not None:
            torch.cuda.current_stream(self.embedding.weight.device).wait_event(ev)

    def state_dict(self, *args, **kwargs):
        self.order_after_grad()
        return super().state_dict(*args, **kwargs)

    @torch.no_grad()
    def load_full_weight(self, full: torch.Tensor):
        lo, hi = self.row_range
        self.order_after_grad()
        self.embedding.weight[: hi - lo].copy_(full[lo:hi])

    @torch.no_grad()
    def full_weight(self) -> torch.Tensor:
        lo, hi = self.row_range
        self.order_after_grad()
        parts = [torch.empty(self.rows_per_rank, self.embed_size, dtype=self.embedding.weight.dtype,
                             device=self.embedding.weight.device) for _ in range(self.world)]
        mine = torch.zeros_like(parts[0])
        mine[: hi - lo] = self.embedding.weight[: hi - lo]
        dist.all_gather(parts, mine, group=self.group)
        return torch.cat(parts, 0)[: self.num_rows]

    def full_state_dict(self):
        return {'embedding.weight': self.full_weight()}


class DenseGradBucket:
    """Data-parallel dense parameters (the MLP / cross / CIN weights, a few MB): their gradients averaged over the ranks
    by ONE all-reduce of a flat, persistent bucket -- bf16 on the wire when every gradient is bf16 (fp32 otherwise) --
    issued on the communication stream, where it queues with the embedding exchanges and overlaps the owner-side reduction
    of the sparse gradient instead of sitting on the compute stream behind a ``torch.cat`` and a copy-back loop.

        bucket = DenseGradBucket(model.parameters(), group)        # once
        loss.backward(); bucket.reduce()                            # every step: pack -> all-reduce -> unpack (averaged)
        bucket.wait()                                               # before the gradients are read (optimizer step)

    Pack and unpack are one multi-tensor launch each (torch._foreach_copy_); the bucket and its per-parameter views are
    allocated once, so the step has a static launch sequence.  A process group of one rank does nothing at all."""

    def __init__(self, params, group=None, wire_dtype: Optional[torch.dtype] = None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group)
        if not self.params:
            raise ValueError("DenseGradBucket: no parameters")
        dev = self.params[0].device
        if wire_dtype is None:
            wire_dtype = torch.bfloat16 if all(p.dtype == torch.bfloat16 for p in self.params) else torch.float32
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=wire_dtype, device=dev)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        self._event = None
        self.bytes_per_step = 0 if self.world == 1 else self.flat.numel() * self.flat.element_size()

    def reduce(self) -> None:
        """average ``p.grad`` over the ranks in place (every parameter must have a gradient)"""
        if self.world == 1:
            return
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("DenseGradBucket.reduce: a parameter has no gradient (run backward first)")
        dev = self.flat.device
        cs = comm_stream(dev)
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream(dev))        # the gradients come first
        with _on_stream(cs), torch.no_grad():
            with _phase("dense all-reduce", dev):
                torch._foreach_copy_(self.views, grads)                 # pack (one launch; casts to the wire dtype)
                dist.all_reduce(self.flat, group=self.group)
                self.flat.mul_(1.0 / self.world)
                torch._foreach_copy_(grads, self.views)                 # unpack
            if cs is not None:
                for g in grads:
                    g.record_stream(cs)
                self._event = torch.cuda.Event()
                self._event.record(cs)

    def wait(self) -> None:
        ev, self._event = self._event, None
        if ev is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(ev)
