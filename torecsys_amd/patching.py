"""``patch(torecsys)``: rebind the reference package's hot-path classes to the HIP drop-ins, so
``torecsys.models.ctr.*`` (which import layers by alias, e.g. ``from torecsys.layers import FMLayer,
DNNLayer`` -- models/ctr/deep_fm.py:6) build on them unchanged.  See INTEGRATION.md.

What is rebound: the interaction layers of SURVEY §8a/§8f-N3, the per-field / deep MLP
(``MultilayerPerceptionLayer`` and its aliases ``DNNLayer``, ``DenseLayer``, ``FullyConnectLayer``,
``FeedForwardLayer`` -- layers/ctr/__init__.py:23-35; this is what ``DeepAndCrossNetworkModel.deep``,
``DeepFactorizationMachineModel.deep`` and ``XDeepFactorizationMachineModel.deep`` are built from,
models/ctr/deep_and_cross_network.py:44, deep_fm.py:47, xdeep_fm.py:71), the three index-embedding
inputs and the ``Inputs`` router (inputs/inputs.py:56-89).  ``patch(pkg, mlp=False)`` /
``patch(pkg, router=False)`` leave the MLP / the router with the reference."""
from __future__ import annotations

import importlib
import sys

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


def patch(torecsys_pkg=None, fuse_fm: bool = True, mlp: bool = True, router: bool = True):
    """Replace the classes in ``torecsys.layers`` / ``torecsys.inputs`` (and in every torecsys module
    that already imported them by name) with the torecsys_amd drop-ins.  Returns the package.

    ``fuse_fm`` (default on): a patched ``MultiIndicesEmbedding`` constructed WITHOUT an explicit ``fuse_fm=``
    produces the FM second-order term inside the lookup kernel and leaves it for ``FMLayer`` (the north-star
    kernel; a model without an FM layer just never reads it: +25 MB of writes at the BASELINE shape).
    ``mlp`` / ``router``: also rebind the MLP layer family / the ``Inputs`` router (module docstring)."""
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
    return torecsys_pkg


def unpatch():
    for (mod_name, n), obj in list(_saved.items()):
        mod = sys.modules.get(mod_name)
        if mod is not None:
            setattr(mod, n, obj)
    _saved.clear()
    if "fuse_fm" in _saved_defaults:
        _inputs.DEFAULT_FUSE_FM = _saved_defaults.pop("fuse_fm")
