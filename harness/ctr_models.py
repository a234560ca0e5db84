"""Test / bench harness (NOT product code): the four callers of the hot path -- M1-M4 of SURVEY.md section 8a.

A torecsys user keeps the reference's own ``torecsys.models.ctr`` classes and swaps the layers underneath them with
``torecsys_amd.patch(torecsys)``.  These stand-ins exist so that ``tests/`` and ``bench.py`` can drive the drop-in
layers on the GPU box, where the reference package is absent.  They keep the reference's constructor keywords and the
attribute names its ``state_dict`` uses (``fm``, ``deep``, ``cross``, ``cin``, ``fc``, ``bias``) so the golden
parameters load by name; everything else is a restatement on plain (un-named) tensors of what each model computes:

  M1  logit = sum_n first[b,n] + sum_e FM(emb)[b,e] (+ bias)                       models/ctr/factorization_machine.py
  M2  logit = sum_n first[b,n] + sum_e FM(emb)[b,e] + MLP(emb as (B, N*E))         models/ctr/deep_fm.py
  M3  logit = Linear([cross(emb) | per-field MLP(emb)] as (B, N*(E+Od)))           models/ctr/deep_and_cross_network.py
  M4  logit = sum_n first[b,n] + CIN(emb) + MLP(emb as (B, N*E)) + bias            models/ctr/xdeep_fm.py

Outputs are (B, 1) un-named tensors, pinned to the reference by ``tests/golden/models.npz``.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from torecsys_amd import functional as F_
from torecsys_amd.layers import CINLayer, CrossNetworkLayer, DNNLayer, FMLayer, strided_outputs


def _plain(t: torch.Tensor) -> torch.Tensor:
    return t.rename(None) if t.has_names() else t


def _first_order(feat: torch.Tensor) -> torch.Tensor:
    """(B, N, 1) first-order weights of the looked-up rows -> (B, 1)"""
    f = _plain(feat)
    return f.reshape(f.shape[0], -1).sum(dim=1, keepdim=True)


FUSED_HEAD = os.environ.get("TRS_FUSED_HEAD", "1") not in ("", "0")


def _fused_head(t: torch.Tensor) -> bool:
    """the models' scalar head (row sums + adds) as one kernel, F_.ctr_logit -- what ``torecsys_amd.patch(pkg, heads=True)``
    puts under the reference's own model classes"""
    return FUSED_HEAD and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16)


def _rows(emb: torch.Tensor) -> torch.Tensor:
    """(B, N, E) block seen as (B, N*E) rows of the sample-wise MLP (a view)"""
    e = _plain(emb)
    return e.reshape(e.shape[0], -1)


class FactorizationMachineModel(nn.Module):
    def __init__(self, use_bias: bool = True, dropout_p: Optional[float] = None):
        super().__init__()
        self.fm = FMLayer(dropout_p)
        self.bias = None
        if use_bias:
            self.bias = nn.Parameter(torch.empty(1, 1).uniform_())

    def forward(self, feat_inputs: torch.Tensor, emb_inputs: torch.Tensor) -> torch.Tensor:
        fm = self.fm(emb_inputs)
        if _fused_head(fm):
            return F_.ctr_logit(fm, feat_inputs, bias=self.bias)
        logit = _plain(fm).sum(dim=1, keepdim=True) + _first_order(feat_inputs)
        return logit if self.bias is None else logit + self.bias


class DeepFactorizationMachineModel(nn.Module):
    def __init__(self, embed_size: int, num_fields: int, deep_layer_sizes: List[int],
                 fm_dropout_p: Optional[float] = None, deep_dropout_p: Optional[List[float]] = None,
                 deep_activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        self.fm = FMLayer(fm_dropout_p)
        self.deep = DNNLayer(inputs_size=num_fields * embed_size, output_size=1, layer_sizes=deep_layer_sizes,
                             dropout_p=deep_dropout_p, activation=deep_activation)

    def forward(self, feat_inputs: torch.Tensor, emb_inputs: torch.Tensor) -> torch.Tensor:
        fm = self.fm(emb_inputs)
        if _fused_head(fm):
            with strided_outputs():          # ctr_logit reads the logit column of the padded output where it lies
                deep = self.deep(_rows(emb_inputs))
        else:
            deep = self.deep(_rows(emb_inputs))
        if _fused_head(fm):
            return F_.ctr_logit(fm, feat_inputs, [deep])
        shallow = _plain(fm).sum(dim=1, keepdim=True) + _first_order(feat_inputs)
        return _plain(deep) + shallow


class DeepAndCrossNetworkModel(nn.Module):
    def __init__(self, inputs_size: int, num_fields: int, deep_output_size: int, deep_layer_sizes: List[int],
                 cross_num_layers: int, output_size: int = 1, deep_dropout_p: Optional[List[float]] = None,
                 deep_activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        self.cross = CrossNetworkLayer(inputs_size=inputs_size, num_layers=cross_num_layers)
        self.deep = DNNLayer(inputs_size=inputs_size, output_size=deep_output_size, layer_sizes=deep_layer_sizes,
                             dropout_p=deep_dropout_p, activation=deep_activation)
        self.fc = nn.Linear(num_fields * (inputs_size + deep_output_size), output_size)

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        crossed = _plain(self.cross(emb_inputs))            # (B, N, E)
        per_field = _plain(self.deep(emb_inputs))           # (B, N, Od): the MLP runs on every field's row
        if _fused_head(crossed) and self.fc.out_features == 1 and F_.cat_head_supported(crossed, per_field, self.fc.weight):
            # the one-output Linear read from the two blocks where they lie: no 0.65 GB concatenation, no slice copies of
            # its gradient (F_.cat_head; TRS_FUSED_HEAD=0 restores the composition below)
            return F_.cat_head(crossed, per_field, self.fc.weight, self.fc.bias)
        both = torch.cat((crossed, per_field), dim=2)       # field-major [cross | deep] rows, as the head expects
        return self.fc(both.reshape(both.shape[0], -1))


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
        self.bias = nn.Parameter(torch.empty(1).uniform_())

    def forward(self, feat_inputs: torch.Tensor, emb_inputs: torch.Tensor) -> torch.Tensor:
        cin = self.cin(emb_inputs)
        if _fused_head(cin):
            with strided_outputs():
                deep = self.deep(_rows(emb_inputs))
        else:
            deep = self.deep(_rows(emb_inputs))
        if _fused_head(cin):
            return F_.ctr_logit(None, feat_inputs, [cin, deep], bias=self.bias)
        wide = _first_order(feat_inputs) + self.bias
        return _plain(cin) + _plain(deep) + wide
