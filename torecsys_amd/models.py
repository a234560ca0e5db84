"""The four caller models of the hot path, composed exactly like the reference's
``torecsys.models.ctr`` classes (same constructor arguments, forward signatures and un-named (B,O)
outputs) from the drop-in layers.  They exist as test / bench harness: a torecsys user keeps the
reference's model classes and swaps the layers (``torecsys_amd.patch``) -- these mirror what then runs.

Reference: models/ctr/factorization_machine.py:42-71, deep_fm.py:55-110,
deep_and_cross_network.py:58-98, xdeep_fm.py:82-124.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .layers import CINLayer, CrossNetworkLayer, DNNLayer, FMLayer


def _named(t: torch.Tensor, names) -> torch.Tensor:
    return t if t.names == tuple(names) else t.rename(None).refine_names(*names)


class FactorizationMachineModel(nn.Module):
    def __init__(self, use_bias: bool = True, dropout_p: Optional[float] = None):
        super().__init__()
        self.fm = FMLayer(dropout_p)
        self.use_bias = use_bias
        if use_bias:
            self.bias = nn.Parameter(torch.zeros((1, 1,), names=('B', 'O',)))
            nn.init.uniform_(self.bias.data)

    def forward(self, feat_inputs: torch.Tensor, emb_inputs: torch.Tensor) -> torch.Tensor:
        feat_inputs = _named(feat_inputs, ('B', 'N', 'E'))
        fm_first = feat_inputs.sum(dim='N').rename(E='O')
        fm_second = self.fm(emb_inputs).sum(dim='O', keepdim=True)
        outputs = fm_second + fm_first
        if self.use_bias:
            outputs = outputs + self.bias
        return outputs.rename(None)


class DeepFactorizationMachineModel(nn.Module):
    def __init__(self, embed_size: int, num_fields: int, deep_layer_sizes: List[int],
                 fm_dropout_p: Optional[float] = None, deep_dropout_p: Optional[List[float]] = None,
                 deep_activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        self.fm = FMLayer(fm_dropout_p)
        self.deep = DNNLayer(inputs_size=num_fields * embed_size, output_size=1, layer_sizes=deep_layer_sizes,
                             dropout_p=deep_dropout_p, activation=deep_activation)

    def forward(self, feat_inputs: torch.Tensor, emb_inputs: torch.Tensor) -> torch.Tensor:
        feat_inputs = _named(feat_inputs, ('B', 'N', 'E'))
        fm_first = feat_inputs.flatten(('N', 'E',), 'O')
        fm_second = self.fm(emb_inputs)
        fm_out = torch.cat([fm_second, fm_first], dim='O')
        fm_out = fm_out.sum(dim='O', keepdim=True)
        emb = _named(emb_inputs, ('B', 'N', 'E'))
        deep_in = emb.flatten(('N', 'E',), 'E')
        deep_out = self.deep(deep_in)
        outputs = deep_out + fm_out
        return outputs.rename(None)


class DeepAndCrossNetworkModel(nn.Module):
    def __init__(self, inputs_size: int, num_fields: int, deep_output_size: int, deep_layer_sizes: List[int],
                 cross_num_layers: int, output_size: int = 1, deep_dropout_p: Optional[List[float]] = None,
                 deep_activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        self.deep = DNNLayer(inputs_size=inputs_size, output_size=deep_output_size, layer_sizes=deep_layer_sizes,
                             dropout_p=deep_dropout_p, activation=deep_activation)
        self.cross = CrossNetworkLayer(inputs_size=inputs_size, num_layers=cross_num_layers)
        cat_size = (deep_output_size + inputs_size) * num_fields
        self.fc = nn.Linear(cat_size, output_size)

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        cross_out = self.cross(emb_inputs)
        deep_out = self.deep(emb_inputs)
        outputs = torch.cat([cross_out, deep_out], dim='O')
        outputs = outputs.flatten(('N', 'O',), 'O')
        outputs = self.fc(outputs.rename(None))
        return outputs


class XDeepFactorizationMachineModel(nn.Module):
    def __init__(self, embed_size: int, num_fields: int, cin_layer_sizes: List[int], deep_layer_sizes: List[int],
                 cin_is_direct: Optional[bool] = False, cin_use_bias: Optional[bool] = True,
                 cin_use_batchnorm: Optional[bool] = True, cin_activation: Optional[nn.Module] = nn.ReLU(),
                 deep_dropout_p: Optional[List[float]] = None, deep_activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        self.cin = CINLayer(embed_size=embed_size, num_fields=num_fields, output_size=1,
                            layer_sizes=cin_layer_sizes, is_direct=cin_is_direct, use_bias=cin_use_bias,
                            use_batchnorm=cin_use_batchnorm, activation=cin_activation)
        self.deep = DNNLayer(inputs_size=embed_size * num_fields, output_size=1, layer_sizes=deep_layer_sizes,
                             dropout_p=deep_dropout_p, activation=deep_activation)
        self.bias = nn.Parameter(torch.zeros(1))
        nn.init.uniform_(self.bias.data)

    def forward(self, feat_inputs: torch.Tensor, emb_inputs: torch.Tensor) -> torch.Tensor:
        feat_inputs = _named(feat_inputs, ('B', 'N', 'E'))
        emb = _named(emb_inputs, ('B', 'N', 'E'))
        deep_inputs = emb.flatten(('N', 'E',), 'E')
        cin_out = self.cin(emb_inputs)
        deep_out = self.deep(deep_inputs)
        feat_output = feat_inputs.sum(dim='N').rename(None).refine_names('B', 'O')
        outputs = feat_output + cin_out + deep_out + self.bias
        return outputs.rename(None)
