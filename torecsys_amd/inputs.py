"""Drop-in index -> embedding input modules (same class names, constructor / forward signatures,
output names and ``state_dict`` keys as ``torecsys.inputs.base``), running on libtrs_hip.so.

Reference: torecsys/inputs/base/{__init__,single_index_emb,multi_indices_emb,
multi_indices_field_aware_emb}.py and torecsys/inputs/inputs.py.  Class ``__name__``s are kept
identical because ``Inputs.forward`` dispatches on them (inputs.py:70,84).
"""
from __future__ import annotations

import os
from collections import namedtuple
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import functional as F_


def _strip(t: torch.Tensor) -> torch.Tensor:
    return t.rename(None) if t.has_names() else t


def _check_embedding_kwargs(kwargs: dict):
    # nn.Embedding options the HIP path does not implement are rejected instead of silently ignored
    if kwargs.get("max_norm") is not None:
        raise NotImplementedError("torecsys_amd: nn.Embedding(max_norm=...) is not supported")
    if kwargs.get("scale_grad_by_freq"):
        raise NotImplementedError("torecsys_amd: nn.Embedding(scale_grad_by_freq=True) is not supported")
    if kwargs.get("sparse"):
        raise NotImplementedError("torecsys_amd: nn.Embedding(sparse=True) is not supported (dense gradient only)")


def field_offsets(field_sizes: List[int]) -> torch.Tensor:
    """(N,) int64 row offsets ``(0, *cumsum(field_sizes)[:-1])`` -- multi_indices_emb.py:54.
    Computed in int64 (the reference goes through float32 and is exact only below 2**24 rows)."""
    sizes = torch.as_tensor(list(field_sizes), dtype=torch.int64)
    return torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)[:-1]])


# What ``MultiIndicesEmbedding(fuse_fm=None)`` resolves to at construction.  ``torecsys_amd.patch(torecsys)`` switches it
# on, so that models built from the reference's own classes run the fused lookup + FM kernel without naming it.
DEFAULT_FUSE_FM = False


class BaseInput(nn.Module):
    """inputs/base/__init__.py:11-45."""

    def __init__(self):
        super().__init__()
        self.schema = None
        self.fused_optimizer = None

    def set_fused_optimizer(self, opt):
        """Apply ``opt`` (torecsys_amd.optim.FusedSparseSGD / FusedSparseAdagrad) to this module's table rows inside
        the backward pass: no dense gradient is produced, ``weight.grad`` stays None.  Returns self."""
        self.fused_optimizer = opt
        return self

    def __len__(self) -> int:
        return self.length

    def set_schema(self, inputs: Union[str, List[str]], **kwargs):
        if isinstance(inputs, str):
            inputs = [inputs]
        schema = namedtuple('Schema', ['inputs'])
        self.schema = schema(inputs=inputs)


class SingleIndexEmbedding(BaseInput):
    """single_index_emb.py:14-59: (B,1)|(B,N) any-int indices -> (B,N,E) named ('B','N','E')."""

    def __init__(self, embed_size: int, field_size: int, padding_idx: Optional[int] = None,
                 nn_embedding: Optional[nn.Parameter] = None, **kwargs):
        super().__init__()
        _check_embedding_kwargs(kwargs)
        if nn_embedding is not None:
            embed_size = nn_embedding.size('E') if nn_embedding.has_names() else nn_embedding.size(-1)
            self.embedding = nn.Embedding.from_pretrained(_strip(nn_embedding))
        else:
            self.embedding = nn.Embedding(field_size, embed_size, padding_idx=padding_idx, **kwargs)
        self.length = embed_size

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        out = F_.gather_rows(self.embedding.weight, inputs, None, self.embedding.padding_idx, self.fused_optimizer)
        out.names = ('B', 'N', 'E',)
        return out


class MultiIndicesEmbedding(BaseInput):
    """multi_indices_emb.py:18-112: one ``sum(field_sizes) x E`` table, per-field offsets,
    (B,N) -> (B,N,E) (or (B,1,N*E) when ``flatten``).

    ``offsets`` is a non-persistent buffer (same ``state_dict`` as the reference -- only
    ``embedding.weight`` -- but it follows ``.to()``).  With ``fuse_fm=True`` (``None`` = the package
    default ``DEFAULT_FUSE_FM``, which ``torecsys_amd.patch()`` turns on) the lookup
    kernel also produces the FM second-order term of the same rows and leaves it on the returned tensor
    for ``FactorizationMachineLayer`` to pick up (one pass over the rows instead of two); ``fuse_ipn=True`` does the
    same for ``InnerProductNetworkLayer`` (bf16 tables the matrix-core pair kernel covers; plain lookup otherwise)."""

    def __init__(self, embed_size: Optional[int] = None, field_sizes: Optional[List[int]] = None,
                 nn_embedding: Optional[nn.Parameter] = None, device: str = 'cpu',
                 flatten: Optional[bool] = False, fuse_fm: Optional[bool] = None, fuse_ipn: bool = False, **kwargs):
        super().__init__()
        _check_embedding_kwargs(kwargs)
        if nn_embedding is not None:
            self.embedding = nn.Embedding.from_pretrained(_strip(nn_embedding))
        elif field_sizes is not None and embed_size is not None:
            self.embedding = nn.Embedding(sum(field_sizes), embed_size, **kwargs)
        else:
            raise ValueError('missing required arguments')
        if field_sizes is None:
            raise ValueError('missing required arguments')
        self.register_buffer('offsets', field_offsets(field_sizes), persistent=False)
        self.flatten = flatten
        # the package default never overrides an explicit request for the fused inner-product lookup, and skips one-wide
        # tables (the first-order ``feat_inputs`` of the CTR models: nobody consumes an FM term of theirs)
        self.fuse_fm = ((DEFAULT_FUSE_FM and not fuse_ipn and self.embedding.embedding_dim > 1) if fuse_fm is None
                        else bool(fuse_fm))
        self.fuse_ipn = fuse_ipn
        self.field_size = self.embedding.num_embeddings
        self.embed_size = self.embedding.embedding_dim
        self.padding_idx = self.embedding.padding_idx
        self.length = self.embed_size * len(field_sizes) if self.flatten else self.embed_size
        self.to(device)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        idx = _strip(inputs)
        if idx.dim() != 2 or idx.shape[1] != self.offsets.numel():
            raise ValueError(f'inputs must be (B, {self.offsets.numel()}), got {tuple(idx.shape)}')
        if self.fuse_fm and not self.flatten:
            out, fm, _ = F_.embed_fm(self.embedding.weight, idx, self.offsets, opt=self.fused_optimizer,
                                     padding_idx=self.padding_idx)
            out._trs_fused_fm = (fm, out._version)
        elif (self.fuse_ipn and not self.flatten and self.padding_idx is None
              and F_.embed_ipn_supported(self.embedding.weight, idx.shape[1])):
            out, ipn = F_.embed_ipn(self.embedding.weight, idx, self.offsets, opt=self.fused_optimizer)
            out._trs_fused_ipn = (ipn, out._version)
        else:
            out = F_.gather_rows(self.embedding.weight, idx, self.offsets, self.padding_idx, self.fused_optimizer)
        if self.flatten:
            out = out.reshape(out.shape[0], 1, -1)
        out.names = ('B', 'N', 'E',)
        return out


class MultiIndicesFieldAwareEmbedding(BaseInput):
    """multi_indices_field_aware_emb.py:24-111: N tables (xavier-uniform), (B,N) -> (B,N*N,E);
    row i*N+j = table i looked up with field j's index."""

    def __init__(self, embed_size: int, field_sizes: List[int], device: str = 'cpu',
                 flatten: Optional[bool] = False):
        super().__init__()
        self.num_fields = len(field_sizes)
        self.embeddings = nn.ModuleList([
            nn.Embedding(sum(field_sizes), embed_size) for _ in range(self.num_fields)
        ])
        for embedding in self.embeddings:
            nn.init.xavier_uniform_(embedding.weight.data)
        self.register_buffer('offsets', field_offsets(field_sizes), persistent=False)
        self.flatten = flatten
        self.length = embed_size
        self.to(device)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        idx = _strip(inputs)
        if idx.dim() != 2 or idx.shape[1] != self.num_fields:
            raise ValueError(f'inputs must be (B, {self.num_fields}), got {tuple(idx.shape)}')
        out = F_.fa_gather_rows([e.weight for e in self.embeddings], idx, self.offsets)
        if self.flatten:
            out = out.reshape(out.shape[0], 1, -1)
        out.names = ('B', 'N', 'E',)
        return out


class ValueInput(BaseInput):
    """inputs/base/value_inp.py:26-44 (pass-through; no kernel)."""

    def __init__(self, num_fields: int, transforms=None):
        super().__init__()
        self.num_fields = num_fields
        self.transforms = transforms
        self.length = 1

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        inputs = _strip(inputs)
        if inputs.dim() == 2:
            inputs = inputs.unsqueeze(dim=-1)
        if self.transforms:
            inputs = self.transforms(inputs)
        inputs.names = ('B', 'N', 'E',)
        return inputs


# TRS_PAIR_FIRST_ORDER=1: Inputs serves the E = 1 first-order table of an FM-family model from the wide table's lookup
# kernel and bucket walk (trs_embed_fm_fields / trs_scatter_rows_first) instead of its own two launches.  Off by default:
# measured on the DeepFM step it is a loss -- the lookup kernel goes from 133 to 162 us and the walk from 176 to 210 us
# (one more dependent 2-byte access per lookup in the two bandwidth-critical kernels) while the separate E = 1 launches
# it removes cost 17 + 28 us: 1.62 vs 1.58 ms per step, and the roofline kernel itself gets slower.
PAIR_FIRST_ORDER = os.environ.get("TRS_PAIR_FIRST_ORDER", "0") == "1"


# 0: every lookup on the caller's stream; 1: lookups beyond the first on the "lookup" side stream, enqueued BEHIND the first
# (the forward stays serial, the backward's bucket walks run side by side); 2: enqueued in FRONT of it (parallel forward)
def _on_hip(t: torch.Tensor) -> bool:
    return t.is_cuda


# a StackedInput of SingleIndexEmbeddings as ONE lookup launch over the separate tables (TRS_STACKED_ONE_LAUNCH=0: its own forward)
STACKED_ONE_LAUNCH = os.environ.get("TRS_STACKED_ONE_LAUNCH", "1") not in ("", "0")
LOOKUP_STREAMS = int(os.environ.get("TRS_LOOKUP_STREAMS", "1") or 0)
_SIDE_LOOKUPS = (SingleIndexEmbedding, MultiIndicesEmbedding, MultiIndicesFieldAwareEmbedding)


class Inputs(BaseInput):
    """Dictionary router, inputs/inputs.py:56-89: for every schema entry gather its named columns,
    ``unsqueeze`` 1-D ones, ``cat`` on dim 1 and call the embedding module.  Integer columns on the HIP device are
    packed by one kernel (trs_pack_columns) instead of N unsqueezes + a cat, and schema entries that name the same
    columns (the E=64 and the E=1 table of one model) receive the SAME index tensor, so the row buckets of the batch
    are built once."""

    def __init__(self, schema: Union[Dict[str, nn.Module], None]):
        super().__init__()
        self.schema = schema if schema is not None else {}
        for k, emb_fn in self.schema.items():
            self.add_module(k, emb_fn)
        self.length = None

    def _first_order_partner(self, key: str) -> Optional[str]:
        """The schema entry whose lookup can ride in ``key``'s kernels: ``key`` is a fuse_fm MultiIndicesEmbedding and
        the partner a MultiIndicesEmbedding(embed_size=1) over the same columns and the same per-field row layout (the
        E-wide table and the first-order weights of one FM-family model).  Decided per call: fused optimizers, dtypes
        and devices can change between steps."""
        a = self.schema[key]
        if not PAIR_FIRST_ORDER:
            return None
        if type(a) is not MultiIndicesEmbedding or not a.fuse_fm or a.flatten or a.padding_idx is not None:
            return None
        if getattr(a, 'fused_optimizer', None) is not None or not a.embedding.weight.is_cuda:
            return None
        wa = a.embedding.weight
        if (wa.shape[1] * wa.element_size()) % 16 != 0:
            return None
        for k, f in self.schema.items():
            if k == key or type(f) is not MultiIndicesEmbedding:
                continue
            if (f.embed_size != 1 or f.flatten or f.fuse_fm or f.padding_idx is not None
                    or getattr(f, 'fused_optimizer', None) is not None):
                continue
            wf = f.embedding.weight
            if (tuple(f.schema.inputs) != tuple(a.schema.inputs) or wf.shape[0] != wa.shape[0] or wf.dtype != wa.dtype
                    or wf.device != wa.device or wf.requires_grad != wa.requires_grad):
                continue
            same = self._same_layout.get((key, k))
            if same is None:      # one device comparison per pair and process (the offsets are construction-time buffers)
                same = f.offsets.numel() == a.offsets.numel() and bool(torch.equal(f.offsets, a.offsets))
                self._same_layout[(key, k)] = same
            if same:
                return k
        return None

    @staticmethod
    def _stacked_single_index_tables(stacked, inputs):
        """(children, their index columns) when ``stacked`` is a StackedInput whose inputs are all plain
        SingleIndexEmbeddings of this package -- one column each, one embed size / dtype / HIP device, no padding row, no
        fused optimizer -- i.e. when F_.gather_rows_tables computes exactly what its forward computes; else None."""
        children = getattr(stacked, 'inputs', None)
        if stacked.__class__.__name__ != 'StackedInput' or not isinstance(children, (list, tuple)) or len(children) < 2:
            return None
        w0 = getattr(getattr(children[0], 'embedding', None), 'weight', None)
        if w0 is None or not _on_hip(w0):
            return None
        raw = []
        for c in children:
            if type(c) is not SingleIndexEmbedding or c.schema is None or len(c.schema.inputs) != 1:
                return None
            w = c.embedding.weight
            if (c.embedding.padding_idx is not None or c.fused_optimizer is not None or w.shape[1] != w0.shape[1]
                    or w.dtype != w0.dtype or w.device != w0.device or not w.is_contiguous()
                    or ((w.shape[1] * w.element_size()) % 16 == 0 and w.data_ptr() % 16 != 0)):
                return None            # (a 16-byte-misaligned table, e.g. a view into a larger buffer: per-child lookups)
            v = inputs[c.schema.inputs[0]]
            v = v.rename(None) if v.has_names() else v
            if v.is_floating_point() or v.device != w0.device or not (v.dim() == 1 or (v.dim() == 2 and v.shape[1] == 1)):
                return None
            raw.append(v)
        if len({v.shape[0] for v in raw}) != 1 or len({v.dtype for v in raw}) != 1 or raw[0].dtype not in (torch.int64, torch.int32):
            return None
        return list(children), raw

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        outputs = {}
        packed = {}                    # tuple of column names -> packed tensor (shared between schema entries)
        if not hasattr(self, '_same_layout'):
            self._same_layout = {}
        partner = {}                   # first-order entry -> the wide entry that computes it
        for k in self.schema:
            p = self._first_order_partner(k)
            if p is not None and p not in partner:
                partner[p] = k
        # pass 1: every entry's index tensor (column packing runs on the caller's stream, once per column set)
        args = {}
        for k, emb_fn in self.schema.items():
            if k in partner and partner[k] in self.schema:
                continue               # a first-order table served by its partner's pass
            if emb_fn.__class__.__name__ in ['ConcatInput', 'StackedInput']:
                stacked = self._stacked_single_index_tables(emb_fn, inputs) if STACKED_ONE_LAUNCH else None
                if stacked is not None:
                    # N SingleIndexEmbeddings under a StackedInput (stacked_inp.py:94-134: N lookups + a cat, and N
                    # bucket builds + walks in the backward): one lookup launch over the N separate tables, one walk
                    children, raw = stacked
                    names = tuple(c.schema.inputs[0] for c in children)
                    idx = packed.get(names)
                    if idx is None:
                        if F_.pack_columns_supported(raw):
                            idx = F_.pack_columns(raw)
                        else:
                            idx = torch.cat([v.unsqueeze(-1) if v.dim() == 1 else v for v in raw], dim=1)
                        packed[names] = idx
                    out = F_.gather_rows_tables([c.embedding.weight for c in children], idx)
                    out.names = ('B', 'N', 'E',)
                    outputs[k] = out
                    continue
                args[k] = [{i: inputs[i] for i in emb_fn.schema.inputs}]
                continue
            names = tuple(emb_fn.schema.inputs)
            inp = packed.get(names)
            if inp is None:
                raw = [inputs[emb_k] for emb_k in names]
                if F_.pack_columns_supported(raw):
                    inp = F_.pack_columns(raw)
                else:
                    cols = [v.unsqueeze(-1) if v.dim() == 1 else v for v in raw]
                    inp = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
                packed[names] = inp
            args[k] = [inp]
        # The lookups of a batch are independent of each other: every plain lookup after the first goes onto the "lookup"
        # side stream.  What pays is the BACKWARD: autograd runs a node's backward on the stream of its forward, so the
        # bucket walk of the E = 1 first-order table of an FM-family model (one walk and four small launches, ~45 us at the
        # end of the step) runs beside the dense backward instead.  In the forward the side lookup is enqueued BEHIND the
        # first one (mode 1): beside it (mode 2) the two gathers share the memory pipe and the wide lookup goes from 131 to
        # 157 us for the 17 us it hides.  Measured alternately on one box, DeepFM step: 0: 1.287 ms, 1: 1.259 ms,
        # 2: 1.304 ms (gpurun_out/r05e, r05f).
        side_keys, first_lookup = [], True
        for k in args:
            emb_fn = self.schema[k]
            plain = (type(emb_fn) in _SIDE_LOOKUPS and k not in partner.values()
                     and isinstance(args[k][0], torch.Tensor) and args[k][0].is_cuda)
            if plain and not first_lookup and LOOKUP_STREAMS:
                side_keys.append(k)
            first_lookup = first_lookup and not plain
        joins = []                     # (event, output) of lookups enqueued on the side stream
        def side_lookups():
            for k in side_keys:
                emb_fn, idx_t = self.schema[k], args[k][0]
                out, ev, side = F_.run_on_side(idx_t.device, "lookup", lambda: emb_fn(idx_t))
                idx_t.record_stream(side)
                joins.append((ev, out))
                outputs[k] = out

        with F_.defer_prefetch():      # row-bucket builds start once every lookup of the batch is enqueued
            if LOOKUP_STREAMS == 2:
                side_lookups()
            for k in args:
                if k in outputs or k in side_keys:
                    continue
                emb_fn, inp_args = self.schema[k], args[k]
                feat_key = next((f for f, w in partner.items() if w == k), None)
                if feat_key is not None:
                    # one lookup pass and one bucket walk for the wide table and its first-order companion
                    feat_fn = self.schema[feat_key]
                    idx = inp_args[0].rename(None) if inp_args[0].has_names() else inp_args[0]
                    if idx.dim() != 2 or idx.shape[1] != emb_fn.offsets.numel():
                        raise ValueError(f'inputs must be (B, {emb_fn.offsets.numel()}), got {tuple(idx.shape)}')
                    out, fm, first = F_.embed_fm_fields(emb_fn.embedding.weight, feat_fn.embedding.weight, idx,
                                                        emb_fn.offsets)
                    out._trs_fused_fm = (fm, out._version)
                    out.names = ('B', 'N', 'E',)
                    first.names = ('B', 'N', 'E',)
                    outputs[k] = out
                    outputs[feat_key] = first
                else:
                    outputs[k] = emb_fn(*inp_args)
            if LOOKUP_STREAMS != 2:
                side_lookups()
        if joins:
            main = F_._abi.current_stream_of(joins[0][1].device)
            for ev, out in joins:      # consumers are enqueued on the caller's stream: it waits for the side lookups here
                main.wait_event(ev)
                extra = [t[0] for t in (getattr(out, a, None) for a in ('_trs_fused_fm', '_trs_fused_ipn')) if t is not None]
                for t in [out] + extra:
                    (t.rename(None) if t.has_names() else t).record_stream(main)
        return {k: outputs[k] for k in self.schema}      # schema order, as the reference returns it

    def add_inputs(self, name: Optional[str] = None, model: Optional[nn.Module] = None,
                   schema: Optional[Dict[str, nn.Module]] = None):
        """Register more input fields after construction: one ``name`` -> ``model`` pair, or a whole ``schema`` dict
        (which wins when both are given).  Same contract as inputs/inputs.py:91-130: TypeError for a non-dict schema,
        a non-str name or a non-Module model; AssertionError for a name that is already routed.  Returns self."""
        if schema is None:
            entries = [(name, model)]
        elif isinstance(schema, dict):
            entries = list(schema.items())
        else:
            raise TypeError(f'schema must be a dict of name -> nn.Module, got {type(schema).__name__}')
        for key, module in entries:
            if not isinstance(key, str):
                raise TypeError(f'input field name must be a str, got {type(key).__name__}')
            if key in self.schema:
                raise AssertionError(f'input field {key!r} is already in the schema')
            if not isinstance(module, nn.Module):
                raise TypeError(f'input field {key!r} must map to an nn.Module, got {type(module).__name__}')
            self.schema[key] = module
            self.add_module(key, module)
        return self
