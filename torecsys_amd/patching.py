"""``patch(torecsys)``: rebind the reference package's hot-path classes to the HIP drop-ins, so
``torecsys.models.ctr.*`` (which import layers by alias, e.g. ``from torecsys.layers import FMLayer,
DNNLayer`` -- models/ctr/deep_fm.py:6) build on them unchanged.  See INTEGRATION.md."""
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
_INPUT_NAMES = ["SingleIndexEmbedding", "MultiIndicesEmbedding", "MultiIndicesFieldAwareEmbedding"]
_saved = {}


def _targets(pkg, names):
    """every already-imported module of ``pkg`` that holds one of ``names`` as an attribute"""
    prefix = pkg.__name__ + "."
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == pkg.__name__ or mod_name.startswith(prefix)):
            continue
        for n in names:
            if n in getattr(mod, "__dict__", {}):
                yield mod, n


def patch(torecsys_pkg=None):
    """Replace the classes in ``torecsys.layers`` / ``torecsys.inputs`` (and in every torecsys module
    that already imported them by name) with the torecsys_amd drop-ins.  Returns the package."""
    if torecsys_pkg is None:
        torecsys_pkg = importlib.import_module("torecsys")
    for names, src in ((_LAYER_NAMES, _layers), (_INPUT_NAMES, _inputs)):
        for mod, n in _targets(torecsys_pkg, names):
            key = (mod.__name__, n)
            if key not in _saved:
                _saved[key] = getattr(mod, n)
            setattr(mod, n, getattr(src, n))
    return torecsys_pkg


def unpatch():
    for (mod_name, n), obj in list(_saved.items()):
        mod = sys.modules.get(mod_name)
        if mod is not None:
            setattr(mod, n, obj)
    _saved.clear()
