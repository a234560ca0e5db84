"""Drop-in feature-interaction layers (same class names, aliases, constructor / forward signatures,
output names and parameter names as ``torecsys.layers.ctr``), running on libtrs_hip.so.

Reference: torecsys/layers/__init__.py (BaseLayer), torecsys/layers/ctr/__init__.py (aliases),
layers/ctr/{factorization_machine,field_aware_factorization_machine,cross_network,
compress_interaction_network,inner_product_network,multilayer_perceptron}.py.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import functional as F_


def _strip(t: torch.Tensor) -> torch.Tensor:
    return t.rename(None) if t.has_names() else t


class BaseLayer(nn.Module, ABC):
    """layers/__init__.py:10-44."""

    def __init__(self, **kwargs):
        super().__init__()

    @property
    @abstractmethod
    def inputs_size(self) -> Dict[str, Tuple[str, ...]]:
        raise NotImplementedError('not implemented')

    @property
    @abstractmethod
    def outputs_size(self) -> Dict[str, Tuple[str, ...]]:
        raise NotImplementedError('not implemented')


class FactorizationMachineLayer(BaseLayer):
    """FM second order, (B,N,E) -> (B,E) named ('B','O'): 0.5*((sum_n x)^2 - sum_n x^2), then dropout.
    layers/ctr/factorization_machine.py:34-81.  ``dropout_p=None`` is treated as 0.0 (the reference's
    model defaults pass None, which makes nn.Dropout raise -- SURVEY §9 Q1)."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'E',)}

    def __init__(self, dropout_p: Optional[float] = 0.0):
        super().__init__()
        self.dropout = nn.Dropout(0.0 if dropout_p is None else dropout_p)

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        fused = getattr(emb_inputs, '_trs_fused_fm', None)
        if fused is not None and fused[1] == emb_inputs._version:
            outputs = fused[0]            # produced by the lookup kernel in the same pass over the rows
        else:
            outputs = F_.fm_layer(_strip(emb_inputs))
        outputs = self.dropout(outputs)
        outputs.names = ('B', 'O',)
        return outputs


class MultilayerPerceptionLayer(BaseLayer):
    """Linear/activation/dropout stack + output Linear.  layers/ctr/multilayer_perceptron.py:24-84.
    Plain GEMMs: stays on nn.Linear (hipBLASLt); outside the hand-written path, inside the timed step."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'N', 'O',)}

    def __init__(self, inputs_size: int, output_size: int, layer_sizes: List[int],
                 dropout_p: Optional[List[float]] = None, activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        if dropout_p is not None and len(dropout_p) != len(layer_sizes):
            raise ValueError('length of dropout_p must be equal to length of layer_sizes.')
        self.embed_size = inputs_size
        layer_sizes = [inputs_size] + layer_sizes
        self.model = nn.Sequential()
        for i, (in_f, out_f) in enumerate(zip(layer_sizes[:-1], layer_sizes[1:])):
            self.model.add_module(f'Linear_{i}', nn.Linear(in_f, out_f))
            if activation is not None:
                self.model.add_module(f'Activation_{i}', activation)
            if dropout_p is not None:
                self.model.add_module(f'Dropout_{i}', nn.Dropout(dropout_p[i]))
        self.model.add_module('LinearOutput', nn.Linear(layer_sizes[-1], output_size))

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        outputs = self.model(_strip(emb_inputs))
        if outputs.dim() == 2:
            outputs.names = ('B', 'O',)
        elif outputs.dim() == 3:
            outputs.names = ('B', 'N', 'O',)
        return outputs


# aliases, layers/ctr/__init__.py:23-35
FMLayer = FactorizationMachineLayer
DenseLayer = MultilayerPerceptionLayer
DNNLayer = MultilayerPerceptionLayer
FullyConnectLayer = MultilayerPerceptionLayer
FeedForwardLayer = MultilayerPerceptionLayer
