"""``patch(torecsys)``: rebind the reference package's hot-path classes to the HIP drop-ins, so
``torecsys.models.ctr.*`` (which import layers by alias, e.g. ``from torecsys.layers import FMLayer,
DNNLayer`` -- models/ctr/deep_fm.py:6) build on them unchanged.  See INTEGRATION.md.

What is rebound: the interaction layers of SURVEY §8a/§8f-N3, the per-field / deep MLP
(``MultilayerPerceptionLayer`` and its aliases ``DNNLayer``, ``DenseLayer``, ``FullyConnectLayer``,
``FeedForwardLayer`` -- layers/ctr/__init__.py:23-35; this is what ``DeepAndCrossNetworkModel.deep``,
``DeepFactorizationMachineModel.deep`` and ``XDeepFactorizationMachineModel.deep`` are built from,
models/ctr/deep_and_cross_network.py:44, deep_fm.py:47, xdeep_fm.py:71), the three index-embedding
inputs and the ``Inputs`` router (inputs/inputs.py:56-89).  ``patch(pkg, mlp=False)`` /
``patch(pkg, router=False)`` leave the MLP / the router with the reference.

``heads`` (default on): the scalar heads of ``FactorizationMachineModel``, ``DeepFactorizationMachineModel`` and
``XDeepFactorizationMachineModel`` (models/ctr/factorization_machine.py:55-66, deep_fm.py:75-104, xdeep_fm.py:117-121:
``cat`` / ``sum('O')`` / ``sum('N')`` / adds on (B, <= 64) tensors, ~20 small ATen launches with their backwards) run as
one kernel, ``functional.ctr_logit``, and the one-output ``fc`` of ``DeepAndCrossNetworkModel``
(deep_and_cross_network.py:82-92: ``cat`` of the cross and deep blocks -> flatten -> Linear) reads the two blocks where
they lie, ``functional.cat_head`` (no 0.65 GB concatenation at the bench size): the classes keep their constructors, parameters and ``state_dict``; only ``forward``
is wrapped, and the wrapper hands anything it does not cover (CPU tensors, dtypes other than fp32 / bf16, models whose
layers are not the drop-ins) to the reference's own ``forward``."""
from __future__ import annotations

import importlib
import sys

import torch

from . import functional as _F
from . import inputs as _inputs
from . import layers as _layers

_LAYER_NAMES = [
    "FactorizationMachineLayer", "FMLayer",
    "FieldAwareFactorizationMachineLayer", "FFMLayer",
    "CrossNetworkLayer",
    "CompressInteractionNetworkLayer", "CINLayer",
    "InnerProductNetworkLayer",
    "OuterProductNetworkLayer",
    "AttentionalFactorizationMachineLayer", "AFMLayer",
    "BilinearInteractionLayer", "FieldAllTypeBilinear", "FieldEachTypeBilinear",
]
_MLP_NAMES = ["MultilayerPerceptionLayer", "DNNLayer", "DenseLayer", "FullyConnectLayer", "FeedForwardLayer"]
_INPUT_NAMES = ["SingleIndexEmbedding", "MultiIndicesEmbedding", "MultiIndicesFieldAwareEmbedding"]
_ROUTER_NAMES = ["Inputs"]
_saved = {}
_saved_defaults = {}


def _targets(pkg, names):
    """every already-imported module of ``pkg`` that holds one of ``names`` as an attribute"""
    prefix = pkg.__name__ + "."
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == pkg.__name__ or mod_name.startswith(prefix)):
            continue
        for n in names:
            if n in getattr(mod, "__dict__", {}):
                yield mod, n


def _plain(t):
    return t.rename(None) if t.has_names() else t


def _head_ok(self, feat_inputs, emb_inputs) -> bool:
    return (torch.is_tensor(feat_inputs) and torch.is_tensor(emb_inputs) and emb_inputs.is_cuda and feat_inputs.is_cuda
            and emb_inputs.dtype == feat_inputs.dtype and emb_inputs.dtype in (torch.float32, torch.bfloat16)
            and emb_inputs.dim() == 3 and feat_inputs.dim() == 3 and feat_inputs.shape[-1] == 1)


def _bias_like(bias, ref):
    """The model's bias (an fp32 nn.Parameter unless the user cast it) in the dtype of the activations: the reference's
    ``outputs += self.bias`` promotes in place, so a bf16 model with an fp32 bias is a working configuration there and
    must stay one here (the cast is differentiable: the gradient arrives in the parameter's own dtype)."""
    b = _plain(bias)
    return b if b.dtype == ref.dtype else b.to(ref.dtype)


def _rows(emb):
    e = _plain(emb)
    return e.reshape(e.shape[0], -1)


def _fm_forward(orig):
    def forward(self, feat_inputs, emb_inputs):
        if not (_head_ok(self, feat_inputs, emb_inputs) and isinstance(self.fm, _layers.FactorizationMachineLayer)):
            return orig(self, feat_inputs, emb_inputs)
        bias = _bias_like(self.bias, emb_inputs) if getattr(self, "use_bias", False) else None
        return _F.ctr_logit(self.fm(emb_inputs), feat_inputs, bias=bias)
    forward._trs_head = True
    return forward


def _deepfm_forward(orig):
    def forward(self, feat_inputs, emb_inputs):
        if not (_head_ok(self, feat_inputs, emb_inputs) and isinstance(self.fm, _layers.FactorizationMachineLayer)
                and isinstance(self.deep, _layers.MultilayerPerceptionLayer)):
            return orig(self, feat_inputs, emb_inputs)
        with _layers.strided_outputs():          # the logit column is read where it lies by ctr_logit
            deep = _plain(self.deep(_rows(emb_inputs)))
        if deep.dtype != emb_inputs.dtype:          # a deep branch kept in another precision: the reference's own arithmetic
            return orig(self, feat_inputs, emb_inputs)
        return _F.ctr_logit(self.fm(emb_inputs), feat_inputs, [deep])
    forward._trs_head = True
    return forward


def _xdeepfm_forward(orig):
    def forward(self, feat_inputs, emb_inputs):
        if not (_head_ok(self, feat_inputs, emb_inputs) and isinstance(self.cin, _layers.CompressInteractionNetworkLayer)
                and isinstance(self.deep, _layers.MultilayerPerceptionLayer)):
            return orig(self, feat_inputs, emb_inputs)
        with _layers.strided_outputs():
            cin, deep = _plain(self.cin(emb_inputs)), _plain(self.deep(_rows(emb_inputs)))  # both (B,1): xdeep_fm.py:60-78
        if cin.dtype != emb_inputs.dtype or deep.dtype != emb_inputs.dtype:
            return orig(self, feat_inputs, emb_inputs)
        return _F.ctr_logit(None, feat_inputs, [cin, deep], bias=_bias_like(self.bias, emb_inputs))
    forward._trs_head = True
    return forward


def _dcn_forward(orig):
    def forward(self, emb_inputs):
        fc = getattr(self, "fc", None)
        if not (torch.is_tensor(emb_inputs) and emb_inputs.is_cuda and emb_inputs.dim() == 3
                and isinstance(self.cross, _layers.CrossNetworkLayer) and isinstance(self.deep, _layers.MultilayerPerceptionLayer)
                and isinstance(fc, torch.nn.Linear) and fc.out_features == 1):
            return orig(self, emb_inputs)
        crossed, per_field = _plain(self.cross(emb_inputs)), _plain(self.deep(emb_inputs))      # (B,N,E), (B,N,Od)
        if not (crossed.dim() == 3 and per_field.dim() == 3 and _F.cat_head_supported(crossed, per_field, fc.weight)):
            both = torch.cat((crossed, per_field), dim=2)          # deep_and_cross_network.py:82-92, un-named
            return fc(both.reshape(both.shape[0], -1))
        return _F.cat_head(crossed, per_field, fc.weight, fc.bias)
    forward._trs_head = True
    return forward


_HEADS = {"FactorizationMachineModel": _fm_forward, "DeepFactorizationMachineModel": _deepfm_forward,
          "XDeepFactorizationMachineModel": _xdeepfm_forward, "DeepAndCrossNetworkModel": _dcn_forward}
_saved_forwards = {}


def _patch_heads(pkg):
    seen = set()
    for mod, n in _targets(pkg, list(_HEADS)):
        cls = getattr(mod, n)
        if id(cls) in seen or not isinstance(cls, type) or getattr(cls.__dict__.get("forward"), "_trs_head", False):
            continue
        seen.add(id(cls))
        orig = cls.forward
        _saved_forwards[cls] = orig
        cls.forward = _HEADS[n](orig)


def patch(torecsys_pkg=None, fuse_fm: bool = True, mlp: bool = True, router: bool = True, heads: bool = True):
    """Replace the classes in ``torecsys.layers`` / ``torecsys.inputs`` (and in every torecsys module
    that already imported them by name) with the torecsys_amd drop-ins.  Returns the package.

    ``fuse_fm`` (default on): a patched ``MultiIndicesEmbedding`` constructed WITHOUT an explicit ``fuse_fm=``
    produces the FM second-order term inside the lookup kernel and leaves it for ``FMLayer`` (the north-star
    kernel; a model without an FM layer just never reads it: +25 MB of writes at the BASELINE shape).
    ``mlp`` / ``router``: also rebind the MLP layer family / the ``Inputs`` router (module docstring).
    ``heads``: wrap the three first-order models' ``forward`` with the one-kernel head (module docstring)."""
    if torecsys_pkg is None:
        torecsys_pkg = importlib.import_module("torecsys")
    groups = [(_LAYER_NAMES, _layers), (_INPUT_NAMES, _inputs)]
    if mlp:
        groups.append((_MLP_NAMES, _layers))
    if router:
        groups.append((_ROUTER_NAMES, _inputs))
    for names, src in groups:
        for mod, n in _targets(torecsys_pkg, names):
            if getattr(mod, n) is getattr(src, n):
                continue
            key = (mod.__name__, n)
            if key not in _saved:
                _saved[key] = getattr(mod, n)
            setattr(mod, n, getattr(src, n))
    if "fuse_fm" not in _saved_defaults:
        _saved_defaults["fuse_fm"] = _inputs.DEFAULT_FUSE_FM
    _inputs.DEFAULT_FUSE_FM = bool(fuse_fm)
    if heads:
        try:
            importlib.import_module(torecsys_pkg.__name__ + ".models")     # the model classes must exist to be wrapped
        except Exception:      # noqa: BLE001 -- a stripped-down package without models: nothing to wrap
            pass
        _patch_heads(torecsys_pkg)
    return torecsys_pkg


def unpatch():
    for (mod_name, n), obj in list(_saved.items()):
        mod = sys.modules.get(mod_name)
        if mod is not None:
            setattr(mod, n, obj)
    _saved.clear()
    for cls, orig in list(_saved_forwards.items()):
        cls.forward = orig
    _saved_forwards.clear()
    if "fuse_fm" in _saved_defaults:
        _inputs.DEFAULT_FUSE_FM = _saved_defaults.pop("fuse_fm")
