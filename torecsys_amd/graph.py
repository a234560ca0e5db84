"""hipGraph capture of one training step.

A CTR step at batch 65 536 is ~110 kernel launches of 5-250 us each; enqueueing them from Python costs ~1.6 ms of
host time, which is the whole budget of the step on an MI355X.  ``GraphedStep`` records the step once (forward,
loss, backward and, when given, the optimizer) into a hipGraph and replays it with one launch; every kernel of
this package is capture safe (no allocation, no host read-back and no synchronisation inside the C ABI; the side
stream that builds the row buckets forks from and joins the capturing stream).

    step = GraphedStep(lambda idx, label: loss_fn(model(**inputs({'c0': idx})), label), (idx0, label0))
    for idx, label in loader:
        loss = step(idx, label)        # copies the batch into the static buffers, replays

The callable must be shape-static and must not branch on tensor values.  Scalars passed by value to kernels (the
learning rate of a fused sparse optimizer) are frozen at capture time; call ``recapture()`` after changing them.
``FusedSparseAdam`` refuses to be captured: its bias-corrected step size changes every step and is a host scalar
(FusedSparseSGD / FusedSparseAdagrad have no per-step host state and capture fine).
The row-sharded multi-GPU lookup is capturable where it reads nothing on the host: one rank, or fixed-capacity slots
(``RowShardedMultiIndicesEmbedding(capacity=...)``: equal all-to-all splits); with exact split sizes it stays eager.
"""
from typing import Callable, Iterable, Optional, Sequence

import warnings

import torch

from . import functional as F_

__all__ = ["GraphedStep", "GraphedRegion"]


class GraphedStep:
    def __init__(self, fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor],
                 params: Optional[Iterable[torch.nn.Parameter]] = None, warmup: int = 3, input_sets: int = 1):
        """``fn(*static_inputs) -> loss`` runs forward AND backward (and optimizer.step() if wanted).
        ``params``: parameters whose ``.grad`` must be dropped before capture so that the captured backward
        allocates them from the graph's private pool (they then stay valid, updated in place, after every replay).
        At least one eager warm-up iteration always runs (on a side stream) before the capture.

        ``input_sets`` = K > 1: K sets of static input buffers, one captured graph per set (all in one memory pool: the
        step's intermediates exist once).  A captured step reads its inputs at fixed addresses, so ``step(*batch)`` has
        to COPY every batch into them first (20 MB of indices at the BASELINE shape: 16 us + a launch gap per 1.25 ms
        step); with K sets the input pipeline writes batch k+1 straight into ``static_inputs((k+1) % K)`` while step k
        runs (its host-to-device copy lands there) and ``replay(k % K)`` copies nothing.  Parameter gradients: every set's
        graph writes its own gradient tensors; ``replay(k)`` rebinds ``p.grad`` to the ones it has just refreshed."""
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedStep: inputs must live on the HIP device (no CPU path)")
        if int(input_sets) < 1:
            raise ValueError("GraphedStep: input_sets must be >= 1")
        self._fn = fn
        self._params = list(params) if params is not None else []
        self._static = [t.clone() for t in example_inputs]
        self._extra_static = [[t.clone() for t in example_inputs] for _ in range(int(input_sets) - 1)]
        self._warmup = int(warmup)
        self._graph = None
        self._extra = []            # (graph, output, [grad per param]) of input sets 1 .. K-1
        self._grads0 = None
        self.output = None
        self.recapture()

    @property
    def input_sets(self) -> int:
        return 1 + len(self._extra_static)

    def static_inputs(self, k: int = 0):
        """the static input tensors of set ``k``: write the next batch into them (in stream order), then ``replay(k)``"""
        return list(self._static if k == 0 else self._extra_static[k - 1])

    def load(self, k: int, *inputs: torch.Tensor) -> None:
        """copy a batch into input set ``k`` (what an input pipeline does instead: produce the batch there)"""
        dsts = self.static_inputs(k)
        if len(inputs) != len(dsts):
            raise ValueError(f"GraphedStep: expected {len(dsts)} inputs, got {len(inputs)}")
        for dst, src in zip(dsts, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"GraphedStep: input {tuple(src.shape)}/{src.dtype} does not match the captured "
                                 f"{tuple(dst.shape)}/{dst.dtype}")
            dst.copy_(src, non_blocking=True)

    def release_outputs(self) -> None:
        """drop the captured outputs (losses with their autograd graphs): needed before ANOTHER capture of the same
        parameters -- a live graph keeps the parameters' AccumulateGrad nodes bound to this capture's stream"""
        self.output = None
        self._extra = [(g, None, grads) for g, _, grads in self._extra]

    def replay(self, k: int = 0) -> torch.Tensor:
        """run the step on whatever input set ``k`` holds: no copy"""
        if k == 0:
            self._graph.replay()
            grads, out = self._grads0, self.output
        else:
            g, out, grads = self._extra[k - 1]
            g.replay()
        if self._extra:
            for p, gr in zip(self._params, grads):
                p.grad = gr
        return out

    def recapture(self):
        # eager warm-up on a side stream: lazy initialisation (hipBLASLt workspaces, TunableOp tuning, kernel
        # attributes) must not happen inside the capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(s):
                for _ in range(max(1, self._warmup)):
                    for p in self._params:
                        p.grad = None
                    self._fn(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for w in caught:
            if "AccumulateGrad node's stream does not match" in str(w.message):
                # the capture would pull the legacy default stream into the graph (hipStreamEndCapture crashes on it)
                raise RuntimeError(
                    "GraphedStep: an autograd graph of an earlier eager step is still alive (a loss or output tensor "
                    "is still referenced), so the parameters' AccumulateGrad nodes are bound to that step's stream. "
                    "Drop those references (del loss) before capturing.")
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        for p in self._params:
            p.grad = None
        # the warm-up left row buckets of the static index buffer in the per-batch cache: a hit would keep their
        # construction (and the event they wait on) OUT of the graph and replay stale buckets for every later batch
        F_.clear_caches()
        self._graph = torch.cuda.CUDAGraph()
        # "thread_local": only THIS thread's calls are policed during the capture.  In the default "global" mode an
        # event query from any other thread -- the watchdog of an initialised RCCL process group polls its events all
        # the time -- invalidates the capture and the runtime aborts the process
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self.output = self._fn(*self._static)
        torch.cuda.synchronize()
        F_.clear_caches()          # entries made during the capture point into the graph's private pool
        self._grads0 = [p.grad for p in self._params]
        self._extra = []
        for st in self._extra_static:      # further input sets: the same step captured on other buffers, same pool
            for p in self._params:
                p.grad = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph.pool(), capture_error_mode="thread_local"):
                out = self._fn(*st)
            torch.cuda.synchronize()
            F_.clear_caches()
            self._extra.append((g, out, [p.grad for p in self._params]))
        if self._extra:
            for p, gr in zip(self._params, self._grads0):
                p.grad = gr

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        if len(inputs) != len(self._static):
            raise ValueError(f"GraphedStep: expected {len(self._static)} inputs, got {len(inputs)}")
        for dst, src in zip(self._static, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"GraphedStep: input {tuple(src.shape)}/{src.dtype} does not match the captured "
                                 f"{tuple(dst.shape)}/{dst.dtype}")
            dst.copy_(src, non_blocking=True)
        return self.replay(0)


class GraphedRegion:
    """hipGraph capture of the DENSE part of a step -- everything between the lookups' outputs and the gradients that
    go back into the lookups -- for steps whose lookups must stay eager (the row-sharded multi-GPU step: its exchanges
    are pipelined across steps on a communication stream, which a whole-step capture would serialise, and RCCL stays out
    of the capture altogether).  The region reads its inputs at FIXED addresses (``RowShardedMultiIndicesEmbedding(
    persistent_outputs=True)`` writes every batch's block / FM term into the same buffers), runs forward + backward and
    leaves

        loss, the gradients w.r.t. the differentiable inputs, and ``p.grad`` of every parameter

    in static tensors.  One replay replaces the ~45 launches of a DeepFM deep branch + head + loss; the host is left with
    the lookups' ~20.

        region = GraphedRegion(lambda block, fm, first, label: crit(model_head(block, fm, first), label),
                               inputs=(block_buf, fm_buf, first_buf, label_buf), diff=(True, True, True, False),
                               params=model.parameters())
        loss, (g_block, g_fm, g_first, _) = region()              # after the eager lookups wrote the buffers
        torch.autograd.backward([block_out, fm_out, first_out], [g_block, g_fm, g_first])    # eager: back to the owners

    ``fn`` receives detached leaves aliasing ``inputs``.  Parameter gradients are OVERWRITTEN by every replay (never
    accumulated): do not set them to None between steps."""

    def __init__(self, fn: Callable[..., torch.Tensor], inputs: Sequence[torch.Tensor], diff: Sequence[bool],
                 params: Optional[Iterable[torch.nn.Parameter]] = None, warmup: int = 2):
        if not all(t.is_cuda for t in inputs):
            raise RuntimeError("GraphedRegion: inputs must live on the HIP device (no CPU path)")
        if len(diff) != len(inputs):
            raise ValueError("GraphedRegion: one ``diff`` flag per input")
        self._fn = fn
        self._inputs = list(inputs)
        self._diff = [bool(d) for d in diff]
        self._params = [p for p in (params or []) if p.requires_grad]
        self._warmup = int(warmup)
        self.loss = None
        self.input_grads = None
        self.recapture()

    def _run(self):
        leaves = [t.detach().requires_grad_() if d else t.detach() for t, d in zip(self._inputs, self._diff)]
        loss = self._fn(*leaves)
        wrt = [t for t, d in zip(leaves, self._diff) if d] + self._params
        grads = list(torch.autograd.grad(loss, wrt, allow_unused=True))
        gin, k = [], 0
        for d in self._diff:
            gin.append(grads[k] if d else None)
            k += 1 if d else 0
        return loss.detach(), gin, grads[k:]

    def recapture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(s):
                for _ in range(max(1, self._warmup)):
                    self._run()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for w in caught:
            if "AccumulateGrad node's stream does not match" in str(w.message):
                # (same hazard as GraphedStep: hipStreamEndCapture crashes when the legacy default stream is pulled in)
                raise RuntimeError(
                    "GraphedRegion: an autograd graph of an earlier eager step is still alive (a loss or output tensor "
                    "is still referenced), so the parameters' AccumulateGrad nodes are bound to that step's stream. "
                    "Drop those references (del loss) before capturing.")
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        F_.clear_caches()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self.loss, self.input_grads, pgrads = self._run()
        torch.cuda.synchronize()
        F_.clear_caches()
        for p, g in zip(self._params, pgrads):
            p.grad = g                     # static tensors of the graph's pool: refreshed in place by every replay

    def __call__(self):
        self._graph.replay()
        return self.loss, self.input_grads
