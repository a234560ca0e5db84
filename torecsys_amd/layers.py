"""Drop-in feature-interaction layers (same class names, aliases, constructor / forward signatures,
output names and parameter names as ``torecsys.layers.ctr``), running on libtrs_hip.so.

Reference: torecsys/layers/__init__.py (BaseLayer), torecsys/layers/ctr/__init__.py (aliases),
layers/ctr/{factorization_machine,field_aware_factorization_machine,cross_network,
compress_interaction_network,inner_product_network,multilayer_perceptron}.py.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple

import math
import os

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


class FieldAwareFactorizationMachineLayer(BaseLayer):
    """FFM pair products, (B,N*N,E) -> (B,NC2,E) named ('B','N','E'):
    out[:,p(i,j)] = x[:,i*N+j] * x[:,j*N+i] for i<j (lexicographic), then dropout.
    layers/ctr/field_aware_factorization_machine.py:36-94."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N^2', 'E',)}

    @property
    def outputs_size(self):
        return {'inputs': ('B', 'NC2', 'E',)}

    def __init__(self, num_fields: int, dropout_p: float = 0.0):
        super().__init__()
        self.num_fields = num_fields
        self.dropout = nn.Dropout(0.0 if dropout_p is None else dropout_p)

    def forward(self, field_emb_inputs: torch.Tensor) -> torch.Tensor:
        outputs = F_.ffm_layer(_strip(field_emb_inputs), self.num_fields)
        outputs = self.dropout(outputs)
        outputs.names = ('B', 'N', 'E',)
        return outputs


class InnerProductNetworkLayer(BaseLayer):
    """Inner-product network, (B,N,E) -> (B,NC2) named ('B','O'):
    out[:,p(i,j)] = sum_e x[:,i,e]*x[:,j,e], i<j.  layers/ctr/inner_product_network.py:34-79.
    (The reference's row_idx/col_idx attributes are not needed: the pair order is computed in-kernel.)"""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'inputs': ('B', 'NC2',)}

    def __init__(self, num_fields: int):
        super().__init__()
        self.num_fields = num_fields
        rows, cols = [], []
        for i in range(num_fields - 1):
            for j in range(i + 1, num_fields):
                rows.append(i)
                cols.append(j)
        self.row_idx = torch.LongTensor(rows)
        self.col_idx = torch.LongTensor(cols)

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        x = _strip(emb_inputs)
        if x.dim() == 3 and x.shape[1] != self.num_fields:
            raise ValueError(f'expected {self.num_fields} fields, got {x.shape[1]}')
        fused = getattr(emb_inputs, '_trs_fused_ipn', None)
        if fused is not None and fused[1] == emb_inputs._version:
            outputs = fused[0]            # produced by the lookup kernel in the same pass over the rows
        else:
            outputs = F_.pair_dot(x)
        outputs.names = ('B', 'O')
        return outputs


def _pair_index_lists(num_fields: int):
    rows, cols = [], []
    for i in range(num_fields - 1):
        for j in range(i + 1, num_fields):
            rows.append(i)
            cols.append(j)
    return torch.LongTensor(rows), torch.LongTensor(cols)


class OuterProductNetworkLayer(BaseLayer):
    """Outer-product network, (B,N,E) -> (B,NC2) named ('B','O').  layers/ctr/outer_product_network.py:36-129.
    ``kernel`` keeps the reference's shape and initialisation (xavier normal, :68-69):
      'mat' (E,NC2,E): out[b,p] = sum_h sum_e x_i[e] kernel[h,p,e] x_j[h]      (:107-121)
      'vec' (1,NC2,E): out[b,p] = sum_e x_i[e] x_j[e] kernel[0,p,e]             (:123-129)
      'num' (1,NC2,1): out[b,p] = kernel[0,p,0] sum_e x_i[e] x_j[e]
    The (B,NC2,E) gathers p, q and the (B,E,NC2,E) product of the 'mat' branch are never formed."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'inputs': ('B', 'NC2',)}

    def __init__(self, embed_size: int, num_fields: int, kernel_type: Optional[str] = 'mat'):
        super().__init__()
        self.row_idx, self.col_idx = _pair_index_lists(num_fields)
        num_pairs = num_fields * (num_fields - 1) // 2
        if kernel_type == 'mat':
            kernel_size = (embed_size, num_pairs, embed_size)
        elif kernel_type == 'vec':
            kernel_size = (1, num_pairs, embed_size)
        elif kernel_type == 'num':
            kernel_size = (1, num_pairs, 1)
        else:
            raise ValueError('kernel_type only allows: ["mat", "num", "vec"].')
        self.kernel_type = kernel_type
        self.embed_size, self.num_fields = embed_size, num_fields
        self.kernel = nn.Parameter(torch.zeros(kernel_size))
        nn.init.xavier_normal_(self.kernel.data)

    def extra_repr(self) -> str:
        return f'kernel_type={self.kernel_type}'

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        x = _strip(emb_inputs)
        if x.dim() != 3 or x.shape[1] != self.num_fields or x.shape[2] != self.embed_size:
            raise ValueError(f'expected (B, {self.num_fields}, {self.embed_size}), got {tuple(x.shape)}')
        if self.kernel_type == 'mat':
            # W[p][e][h] = kernel[h,p,e]: the per-pair matrix applied to x_i (autograd routes the gradient back)
            outputs = F_.pair_bilinear(x, self.kernel.permute(1, 2, 0), None, 0)
        elif self.kernel_type == 'vec':
            outputs = F_.opn_vec(x, self.kernel[0], False)
        else:
            # one scalar per pair: the inner-product kernel (MFMA Gram matrix per sample) times the kernel row
            outputs = F_.pair_dot(x) * self.kernel[0, :, 0].to(x.dtype)
        outputs.names = ('B', 'O')
        return outputs


class FieldAllTypeBilinear(BaseLayer):
    """``y = (x1 @ W) * x2 + b`` with one (E,E) matrix for every pair.  bilinear_interaction.py:11-80."""

    @property
    def inputs_size(self):
        return {'inputs1': ('B', 'NC2', 'E',), 'inputs2': ('B', 'NC2', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'NC2', 'E',)}

    def __init__(self, in1_features, in2_features, bias=True):
        super().__init__()
        self.in1_features, self.in2_features = in1_features, in2_features
        self.weight = nn.Parameter(torch.Tensor(in1_features, in2_features))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(in2_features))
        else:
            # the reference registers nn.Parameter(torch.tensor([0])) here (:62): an int64 tensor cannot require grad
            raise RuntimeError('Only Tensors of floating point and complex dtype can require gradients '
                               '(bias=False cannot be constructed in the reference either)')
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1 / math.sqrt(self.weight.shape[0])
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input1, input2):
        """(B,P,E) pair operands as ``BilinearInteractionLayer`` of the reference hands them over (:241-249; the drop-in
        ``BilinearInteractionLayer`` above never materialises them and does not come through here): one library GEMM
        with the shared matrix, then the product + bias pass on HIP.  Raises off-GPU like every module of this package."""
        x1, x2 = _strip(input1), _strip(input2)
        F_.require_device(x1, x2, self.weight)
        return F_.rows_mul_bias(torch.matmul(x1, self.weight.to(x1.dtype)), x2, self.bias, False)

    def extra_repr(self):
        return f'{self.in1_features} x {self.in2_features} shared by every pair, bias={self.bias is not None}'


class FieldEachTypeBilinear(BaseLayer):
    """``y[:,p] = (x1[:,p] @ W[p]) * x2[:,p] + b[p]``, one matrix per pair.  bilinear_interaction.py:82-152."""

    @property
    def inputs_size(self):
        return {'inputs1': ('B', 'NC2', 'E',), 'inputs2': ('B', 'NC2', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'NC2', 'E',)}

    def __init__(self, in_features, in1_features, in2_features, bias=True):
        super().__init__()
        self.in1_features, self.in2_features = in1_features, in2_features
        self.weight = nn.Parameter(torch.Tensor(in_features, in1_features, in2_features))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(in_features, in2_features))
        else:
            raise RuntimeError('Only Tensors of floating point and complex dtype can require gradients '
                               '(bias=False cannot be constructed in the reference either)')
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1 / math.sqrt(self.weight.shape[0])
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input1, input2):
        """(B,P,E) pair operands (see FieldAllTypeBilinear.forward): a batched library GEMM over the P matrices
        ((P,B,E) @ (P,E,E)), then the product + per-pair bias pass on HIP."""
        x1, x2 = _strip(input1), _strip(input2)
        F_.require_device(x1, x2, self.weight)
        if x1.dim() < 2 or x1.shape[:-1] != x2.shape[:-1]:
            raise ValueError(f'FieldEachTypeBilinear operands must be (*, P, in1) and (*, P, in2), got '
                             f'{tuple(x1.shape)} and {tuple(x2.shape)}')
        P = x1.shape[-2]
        lead = x1.shape[:-2]
        # leading dimensions folded into the batch axis; in1 != in2 is allowed as in the reference (:143-148): the
        # product pass compares T (.., P, in2) with x2
        a3, c3 = x1.reshape(-1, P, x1.shape[-1]), x2.reshape(-1, P, x2.shape[-1])
        T = torch.bmm(a3.transpose(0, 1), self.weight.to(x1.dtype)).transpose(0, 1)
        return F_.rows_mul_bias(T, c3, self.bias, True).reshape(*lead, P, self.in2_features)

    def extra_repr(self):
        return (f'{self.weight.shape[0]} pair matrices of {self.in1_features} x {self.in2_features}, '
                f'bias={self.bias is not None}')


class BilinearInteractionLayer(BaseLayer):
    """Bilinear interaction (FiBiNET), (B,N,E) -> (B,NC2,E) named ('B','N','O').  bilinear_interaction.py:155-255.
    'all':  out[b,p,:] = (x_i @ W) * x_j + b    -- x @ W is ONE (B*N,E)x(E,E) GEMM, the pair products one HIP pass
    'each': out[b,p,:] = (x_i @ W[p]) * x_j + b[p]
    Parameters ``bilinear.weight`` / ``bilinear.bias`` as in the reference; 'interaction' is NotImplemented there
    (:214) and here; bias=False raises as it does there."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'NC2', 'E',)}

    def __init__(self, embed_size: int, num_fields: int, bilinear_type: str = 'all', bias: bool = True):
        super().__init__()
        self.row_idx, self.col_idx = _pair_index_lists(num_fields)
        num_interaction = num_fields * (num_fields - 1) // 2
        self.bilinear_type = bilinear_type
        self.embed_size, self.num_fields = embed_size, num_fields
        if bilinear_type == 'all':
            self.bilinear = FieldAllTypeBilinear(embed_size, embed_size, bias=bias)
        elif bilinear_type == 'each':
            self.bilinear = FieldEachTypeBilinear(num_interaction, embed_size, embed_size, bias=bias)
        elif bilinear_type == 'interaction':
            raise NotImplementedError()
        else:
            raise ValueError('bilinear_type only allows: ["all", "each", "interaction"].')

    def extra_repr(self) -> str:
        return f'bilinear_type={self.bilinear_type}'

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        x = _strip(emb_inputs)
        if x.dim() != 3 or x.shape[1] != self.num_fields or x.shape[2] != self.embed_size:
            raise ValueError(f'expected (B, {self.num_fields}, {self.embed_size}), got {tuple(x.shape)}')
        if self.bilinear_type == 'all':
            W = self.bilinear.weight
            if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and torch.is_grad_enabled():
                # x @ W with the split-K weight gradient of the MLP stack (K = B*N rows against an E x E output)
                T = _LinearSplitK.apply(x, W.t(), None, False, None, None)
            else:
                T = torch.matmul(x, W)
            output = F_.pair_mul(T, x, self.bilinear.bias, False)
        else:
            output = F_.pair_bilinear(x, self.bilinear.weight, self.bilinear.bias, 1)
        output.names = ('B', 'N', 'O',)
        return output


class AttentionalFactorizationMachineLayer(BaseLayer):
    """Attentional FM, (B,N,E) -> ((B,E) named ('B','E'), (B,NC2,1) un-named attention scores).
    layers/ctr/attentional_factorization_machine.py:49-125.  Parameters as in the reference:
    ``attention.Linear.{weight (A,E), bias}``, ``attention.OutProj.{weight (1,A), bias}``; the Softmax / Dropout entries
    of ``attention`` and the output ``dropout`` are kept as modules.  The score dropout (``attention.Dropout``, p = 0.1 by
    default) acts BEFORE the weighted sum in the reference (:82, :105-113); in training the fused kernel takes a keep mask
    drawn here and applies it to the scores it returns and to the sum (trs_afm_fwd_dropout / trs_afm_bwd_dropout)."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'E',), 'attn_scores': ('B', 'NC2', '1',)}

    def __init__(self, embed_size: int, num_fields: int, attn_size: int, dropout_p: float = 0.1):
        super().__init__()
        self.row_idx, self.col_idx = _pair_index_lists(num_fields)
        self.embed_size, self.num_fields = embed_size, num_fields
        self.attention = nn.Sequential()
        self.attention.add_module('Linear', nn.Linear(embed_size, attn_size))
        self.attention.add_module('Activation', nn.ReLU())
        self.attention.add_module('OutProj', nn.Linear(attn_size, 1))
        self.attention.add_module('Softmax', nn.Softmax(dim=1))
        self.attention.add_module('Dropout', nn.Dropout(dropout_p))
        self.dropout = nn.Dropout(dropout_p)

    def forward(self, emb_inputs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = _strip(emb_inputs)
        if x.dim() != 3 or x.shape[1] != self.num_fields or x.shape[2] != self.embed_size:
            raise ValueError(f'expected (B, {self.num_fields}, {self.embed_size}), got {tuple(x.shape)}')
        lin, proj, drop = self.attention.Linear, self.attention.OutProj, self.attention.Dropout
        keep, scale = None, 1.0
        if drop.training and drop.p > 0.0:
            # the dropout the reference applies to the scores inside ``self.attention`` (:82): only the random bits
            # come from torch's generator (as nn.Dropout's do); masking, rescaling and the sum happen in the kernel
            P = self.num_fields * (self.num_fields - 1) // 2
            keep = torch.empty(x.shape[0], P, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - drop.p)
            scale = 1.0 / (1.0 - drop.p) if drop.p < 1.0 else 0.0
        outputs, attn = F_.afm(x, lin.weight, lin.bias, proj.weight, proj.bias, keep, scale)
        attn_scores = attn.unsqueeze(-1)
        outputs.names = ('B', 'E')
        outputs = self.dropout(outputs)
        return outputs, attn_scores


class CrossNetworkLayer(BaseLayer):
    """Cross network, (B,N,E) -> (B,N,E) named ('B','N','O'): x_{l+1} = x0 * (x_l W_l^T + b_l) + x0.
    layers/ctr/cross_network.py:34-87.  Parameters ``model.{l}.weight`` (E,E) / ``model.{l}.bias`` (E)
    as in the reference (a ModuleList of nn.Linear).  Reproduced quirks: the residual adds x0, W_l is a
    full ExE matrix, and the running value starts from ``emb_inputs.detach()`` (:65) so no gradient
    flows through layer 0's linear input (``faithful_grad=False`` gives the textbook gradient).
    Not reproduced: the in-place un-naming of the caller's tensor (:68).  Like the reference, only
    3-D inputs work (its einsum at :78 raises RuntimeError on 2-D inputs)."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'N', 'E',)}

    def __init__(self, inputs_size: int, num_layers: int, faithful_grad: bool = True):
        super().__init__()
        self.embed_size = inputs_size
        self.faithful_grad = faithful_grad
        self.model = nn.ModuleList()
        for _ in range(num_layers):
            self.model.append(nn.Linear(inputs_size, inputs_size))

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        x = _strip(emb_inputs)
        if x.dim() != 3:
            raise RuntimeError(f'CrossNetworkLayer expects a (B, N, E) tensor, got {x.dim()}-D '
                               '(the reference einsum(\'ijk,ijk->ijk\') raises for other ranks)')
        if len(self.model) == 0:
            outputs = x.detach() if self.faithful_grad else x
        else:
            W = torch.stack([layer.weight for layer in self.model])
            b = torch.stack([layer.bias for layer in self.model])
            outputs = F_.cross_network(x, W, b, self.faithful_grad)
        outputs.names = ('B', 'N', 'O',)
        return outputs


class CompressInteractionNetworkLayer(BaseLayer):
    """Compress Interaction Network, (B,N,E) -> (B,O) named ('B','O').
    layers/ctr/compress_interaction_network.py:37-184.  Same module tree as the reference
    (``model.{i}.Conv1d`` / ``.Batchnorm`` / ``.Activation``, ``fc``) so checkpoints load unchanged.
    Per layer the (B,N*H,E) outer product of the reference (:125-132) is never built: the HIP kernel
    forms it on the fly and contracts it with the Conv1d(k=1) weight; BatchNorm1d / activation are the
    layer's own nn.Modules applied to the (B,C,E) result (batch statistics, running stats and any
    activation therefore behave exactly as in the reference).  Reproduced quirk: unless ``is_direct``
    EVERY layer has 2*H output channels and is split by ``chunk`` (the guards at :69/:151 never fire)."""

    @property
    def inputs_size(self):
        return {'inputs': ('B', 'N', 'E',)}

    @property
    def outputs_size(self):
        return {'outputs': ('B', 'O',)}

    def __init__(self, embed_size: int, num_fields: int, output_size: int, layer_sizes: List[int],
                 is_direct: bool = False, use_bias: bool = True, use_batchnorm: bool = True,
                 activation: Optional[nn.Module] = nn.ReLU()):
        super().__init__()
        self.embed_size = embed_size
        self.is_direct = is_direct
        self.layer_sizes = [num_fields] + layer_sizes
        self.model = nn.ModuleList()
        for i, (s_i, s_j) in enumerate(zip(self.layer_sizes[:-1], self.layer_sizes[1:])):
            in_c = self.layer_sizes[0] * s_i
            out_c = s_j if is_direct or i == (len(self.layer_sizes) - 1) else s_j * 2
            cin = nn.Sequential()
            cin.add_module('Conv1d', nn.Conv1d(in_c, out_c, kernel_size=1, bias=use_bias))
            if use_batchnorm:
                cin.add_module('Batchnorm', nn.BatchNorm1d(out_c))
            if activation is not None:
                cin.add_module('Activation', activation)
            self.model.append(cin)
        self.fc = nn.Linear(int(sum(layer_sizes)), output_size)

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        x0 = _strip(emb_inputs)
        if x0.dim() != 3:
            raise ValueError(f'CIN input must be (B, N, E), got {tuple(x0.shape)}')
        outs = [seq.Conv1d.out_channels for seq in self.model]
        hiddens = [c if self.is_direct else c // 2 for c in outs[:-1]]      # widths fed to layers 1..
        if F_.cin_cl_supported(x0, outs, hiddens) and (x0.shape[1] + 31) // 32 in (1, 2, 4, 8):
            return self._forward_channels_last(x0)
        hidden = x0                                   # (B,H,E) channels-first, H_0 = N
        direct_list = []
        for seq in self.model:
            conv = seq.Conv1d
            y = F_.cin_contract(x0, hidden, conv.weight.squeeze(-1), conv.bias)      # (B,C,E)
            for name, mod in seq.named_children():
                if name != 'Conv1d':
                    y = mod(y)
            if self.is_direct:
                direct, hidden = y, y
            else:
                direct, hidden = torch.chunk(y, 2, dim=1)
            direct_list.append(direct)
        pooled = torch.cat(direct_list, dim=1).sum(dim=-1)
        outputs = self.fc(pooled)
        outputs.names = ('B', 'O',)
        return outputs

    def _forward_channels_last(self, x0: torch.Tensor) -> torch.Tensor:
        """bf16 MFMA path: activations are kept channels-last, (B,E,C) -- which is the reference's own
        ``align_to('B','E','N')`` orientation (:105) -- so every layer's output is directly the next layer's
        matrix-core operand; BatchNorm1d sees the (B*E, C) view (same per-channel statistics over (B,E))."""
        B, N, E = x0.shape
        ld0 = ((N + 31) // 32) * 32
        if TRANSPOSE_PAD and F_.transpose_pad_supported(x0, ld0):
            x0T = F_.transpose_pad(x0, ld0)              # one pass; its backward is the transposition back
        else:
            x0T = x0.new_zeros(B, E, ld0)
            x0T[:, :, :N] = x0.transpose(1, 2)
        hiddenT, H = x0T, N
        pooled = []
        n_layers = len(self.model)
        for li, seq in enumerate(self.model):
            conv = seq.Conv1d
            C = conv.out_channels
            D, Hs = (C, 0) if self.is_direct else (C // 2, C // 2)
            rest = [(name, mod) for name, mod in seq.named_children() if name != 'Conv1d']
            names = [name for name, _ in rest]
            per_channel = (names in (['Batchnorm', 'Activation'], ['Activation']) and type(rest[-1][1]) is nn.ReLU
                           and (len(rest) == 1 or type(rest[0][1]) is nn.BatchNorm1d))
            # The LAST layer's "hidden" half (channels [D, C)) is split off and never used (the reference does the same:
            # compress_interaction_network.py:151-156 splits every layer, :176-181 reads the direct halves only): its
            # gradient is exactly zero, through BatchNorm1d and ReLU too (both act per channel), and the contraction's
            # backward is told so.  The forward still computes those channels: BatchNorm's running statistics of them are
            # part of the module's state.
            live = D if (li == n_layers - 1 and not self.is_direct and per_channel) else None
            yT = F_.cin_contract_cl(x0T, hiddenT, conv.weight.squeeze(-1), conv.bias, N, H,
                                    x0_cf=x0 if x0.is_contiguous() else None,
                                    xk_cf=getattr(hiddenT, '_trs_cf', None), live=live)          # (B,E,C)
            fusable = per_channel and F_.cin_glue_supported(yT, D, Hs)
            if fusable:
                # BatchNorm1d + ReLU + chunk + the sum over E of the direct half in two HIP passes (trs_cin_glue_*).
                # Last layer: its hidden half feeds nothing, so the pass does not write it out (Hs = C: no hidden
                # channels; the statistics pass still covers every channel)
                hs_out = C if live is not None else Hs
                hiddenT, pool = F_.cin_glue(yT, rest[0][1] if len(rest) == 2 else None, D, hs_out)
                H = C - Hs
                pooled.append(pool)
                continue
            y2 = yT.reshape(B * E, C)
            for name, mod in rest:
                y2 = mod(y2)
            yT = y2.reshape(B, E, C)
            if self.is_direct:
                directT, hiddenT, H = yT, yT, C
            else:
                directT, hiddenT = torch.chunk(yT, 2, dim=2)
                H = C // 2
            pooled.append(directT.sum(dim=1))
        outputs = self.fc(torch.cat(pooled, dim=1))
        outputs.names = ('B', 'O',)
        return outputs


TRANSPOSE_PAD = os.environ.get("TRS_TRANSPOSE_PAD", "1") not in ("", "0")   # CIN entry: (B,N,E) -> (B,E,ld0) in one pass
PAD_MULTIPLE = int(os.environ.get("TRS_PAD_MULTIPLE", "128"))        # hidden widths are zero-padded to a multiple of this inside the GEMMs
PAD_MIN_WIDTH = 192
PAD_MIN_ROWS = 4096
HYBRID_ONE_NODE = os.environ.get("TRS_HYBRID_ONE_NODE", "1") not in ("", "0")
HYBRID_MLP = os.environ.get("TRS_HYBRID_MLP", "1") not in ("", "0")   # fused tail behind a wide first layer
# _HybridMLP's weight copies into fragment order: 0 (default) in front of each kernel; 1 one launch on the "pack" side stream
# beside the first GEMM; 2 one launch on the caller's stream in front of the first GEMM.  Measured alternately on one box
# (gpurun_out/r05e, r05f; DeepFM step replayed from a hipGraph): 0: 1.259 ms, 2: 1.279 ms, 1: 1.277 ms -- three launches
# fewer on the critical path and still 20 us slower (what the graph's branches overlap with moves: the bucket build no
# longer runs beside the first GEMM).  Kept as a switch and as the ABI's PACK / RUN phases; off.
HOIST_PACK = int(os.environ.get("TRS_HOIST_PACK", "0") or 0)
# _HybridMLP.backward: the tail's weight-gradient launches on a side stream beside the first layer's gradient GEMMs.
# Measured alternately on one box (gpurun_out/r05p): 1.1956 ms off, 1.1975 ms on -- matrix-core kernels beside matrix-core
# kernels only move time.  Off.
WGRAD_STREAM = os.environ.get("TRS_WGRAD_STREAM", "0") not in ("", "0")
# _HybridMLP.backward: the two square tail layers' weight gradients as half-size launches on two streams (see tail_grads)
WGRAD_PAIR = os.environ.get("TRS_WGRAD_PAIR", "0") not in ("", "0")
# _dense_layer_grads: the input gradient behind the weight gradient (see there)
GX_LAST = os.environ.get("TRS_GX_LAST", "1") not in ("", "0")
# the fused tail's forward on the first 416 columns of the 512-wide first-layer output, by the row-owner kernel (mixed family)
MIXED_TAIL = os.environ.get("TRS_MIXED_TAIL", "1") not in ("", "0")


def _pad_width(width: int) -> int:
    """Width the GEMMs run at: hipBLASLt's macro tiles are 128/256 wide, a 400-wide layer runs 1.5-1.6x slower
    than a zero-padded 512-wide one (measured, tools/mlp_pad_probe.py) although it does 28 % more flops."""
    if width < PAD_MIN_WIDTH or width % PAD_MULTIPLE == 0:
        return width
    return (width + PAD_MULTIPLE - 1) // PAD_MULTIPLE * PAD_MULTIPLE


class _PaddedLinear:
    """Zero-padded copies of one nn.Linear's weight/bias, refreshed on every training forward (parameters the user owns
    can change without any version bump) and on every hipGraph capture, where the refresh must be part of the replayed
    work.  The parameters themselves keep the reference's shapes.  ``get_many`` refreshes the copies of a whole stack
    with ONE launch (trs_copy_padded_many) instead of two small copies per layer."""
    __slots__ = ("w", "b", "key")
    _desc_cache = {}

    @staticmethod
    def _state(mod: nn.Linear, in_pad: int, out_pad: int):
        w, b = mod.weight, mod.bias
        slots = mod.__dict__.get('_trs_padded')      # (out_pad, in_pad) -> copy: a layer can be wanted at two widths (the
        if slots is None:                            # library GEMM behind it at 512, the fused tail's forward at 416)
            slots = mod.__dict__['_trs_padded'] = {}
        st = slots.get((out_pad, in_pad))
        if st is None or st.w.dtype != w.dtype or st.w.device != w.device:
            if len(slots) >= 4:
                slots.clear()
            st = _PaddedLinear()
            st.w = torch.zeros(out_pad, in_pad, dtype=w.dtype, device=w.device)
            st.b = torch.zeros(out_pad, dtype=w.dtype, device=w.device) if b is not None else None
            st.key = None
            slots[(out_pad, in_pad)] = st
        return st

    @staticmethod
    def _stale(mod: nn.Linear, st) -> bool:
        # refreshed on EVERY forward that can be followed by a parameter update (grad mode): in-place writes through
        # ``p.data`` (p.data.add_(), clipping, EMA swap-in, dist.broadcast(p.data)) do not bump ``_version``, so a
        # version key would leave the padded copy stale.  Inference (no_grad) keeps the (ptr, version) key.
        w, b = mod.weight, mod.bias
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
        stale = st.key != key or torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing()
        st.key = key
        return stale

    @staticmethod
    def get(mod: nn.Linear, in_pad: int, out_pad: int):
        st = _PaddedLinear._state(mod, in_pad, out_pad)
        if _PaddedLinear._stale(mod, st):
            w, b = mod.weight, mod.bias
            with torch.no_grad():
                st.w[:w.shape[0], :w.shape[1]].copy_(w)
                if b is not None:
                    st.b[:b.shape[0]].copy_(b)
        return st.w, st.b

    @staticmethod
    def get_many(items):
        """items: [(nn.Linear, in_pad, out_pad)] -> [(w_pad, b_pad)], stale copies refreshed by one kernel."""
        states = [_PaddedLinear._state(m, i, o) for m, i, o in items]
        jobs = []
        for (mod, _, _), st in zip(items, states):
            if _PaddedLinear._stale(mod, st):
                w, b = mod.weight, mod.bias
                if not (w.is_cuda and w.is_contiguous() and (b is None or b.is_contiguous())):
                    with torch.no_grad():
                        st.w[:w.shape[0], :w.shape[1]].copy_(w)
                        if b is not None:
                            st.b[:b.shape[0]].copy_(b)
                    continue
                jobs.append((w.data_ptr(), st.w.data_ptr(), w.shape[0], w.shape[1], w.shape[1], st.w.shape[1]))
                if b is not None:
                    jobs.append((b.data_ptr(), st.b.data_ptr(), 1, b.shape[0], b.shape[0], st.b.shape[0]))
        if jobs:
            w0 = items[0][0].weight
            key = (str(w0.device), *jobs)     # addresses repeat across the GPUs of one process: the device is part of the key
            desc = _PaddedLinear._desc_cache.get(key)
            if desc is None:      # a blocking 48-byte-per-copy upload, once per set of parameter storages
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("torecsys_amd: the padded-weight copy descriptors of this MLP are not on the device "
                                       "yet (a host-to-device upload cannot run inside a hipGraph capture); run one "
                                       "eager step before capturing (graph.GraphedStep warms up by default)")
                if len(_PaddedLinear._desc_cache) > 64:
                    _PaddedLinear._desc_cache.clear()
                desc = torch.tensor(jobs, dtype=torch.int64, device=w0.device)
                _PaddedLinear._desc_cache[key] = desc
            F_.call("trs_copy_padded_many", F_.ptr(desc), len(jobs), w0.element_size(), max(j[2] * j[3] for j in jobs),
                    F_.stream_ptr())
        return [(st.w, st.b) for st in states]


def _split_count(rows: int, slice_rows: int, max_slices: int = 384) -> int:
    """Number of split-K slices of a weight gradient over ``rows``: slices of ``slice_rows`` rows, but no more than
    ``max_slices`` of them (at 2.5 M rows -- DCN's per-field MLP -- 1 248 slices of 2 048 rows write and re-read 1.3 GB
    of fp32 partials per layer and run no faster than 312-384 slices: tools/wgrad_probe4.py); 0 = do not split."""
    s = min(rows // slice_rows, max_slices)
    while s >= 4 and rows % s:
        s -= 1
    return s if s >= 4 else 0


class _LinearSplitK(torch.autograd.Function):
    """``F.linear`` (+ optional ReLU) for the MLP stack on the HIP device, plain PyTorch / hipBLASLt GEMMs arranged
    for this shape class (tens of thousands of rows against a few hundred features):
      * forward: bias and ReLU ride in the GEMM epilogue (``torch._addmm_activation``) instead of a second pass;
      * the GEMMs can run on zero-padded copies of the weights (``w_pad``/``b_pad``, see _PaddedLinear): the padded
        output columns are exactly 0 and feed zero weight columns of the next layer, so results are unchanged;
      * weight gradient ``g^T x`` (K = the batch) as a split-K batched GEMM with fp32 partial sums: one GEMM with
        such a deep K and a tiny output runs at ~120 TFLOP/s in hipBLASLt, 32 slices of 2 048 rows at ~370;
      * ReLU backward + bias gradient in one HIP pass (trs_relu_bwd_bias).
    Gradients are returned for the un-padded parameters (a strided slice folded into the final cast)."""

    SPLIT_ROWS = 2048

    @staticmethod
    def forward(ctx, x, weight, bias, fuse_relu, w_pad, b_pad):
        W = weight if w_pad is None else w_pad
        Bv = bias if w_pad is None else b_pad
        x2 = x.reshape(-1, x.shape[-1])
        if fuse_relu and Bv is not None and x2.is_cuda:
            y = torch._addmm_activation(Bv, x2, W.t(), use_gelu=False).reshape(*x.shape[:-1], W.shape[0])
        else:
            y = torch.nn.functional.linear(x, W, Bv)
            if fuse_relu:
                y = torch.relu_(y)
        ctx.has_bias = bias is not None
        ctx.fuse_relu = bool(fuse_relu) and bias is not None and F_.relu_bwd_bias_supported(y.reshape(-1, y.shape[-1]))
        ctx.relu = bool(fuse_relu)
        ctx.wshape = tuple(weight.shape)
        ctx.wdtype = weight.dtype
        ctx.save_for_backward(x, W, y if fuse_relu else None)      # relu output: its sign pattern is the backward mask
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        out_f, in_f = ctx.wshape
        gx = gw = gb = None
        if ctx.fuse_relu:
            # one HIP pass: relu backward + bias gradient (ATen: threshold_backward, then a column sum re-reading it)
            gz, gbf = F_.relu_bwd_bias(g.reshape(-1, g.shape[-1]), y.reshape(-1, y.shape[-1]))
            g = gz.reshape(g.shape)
            gb = gbf[:out_f].to(ctx.wdtype)
        elif ctx.relu:
            g = g * (y > 0).to(g.dtype)
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        if ctx.needs_input_grad[0]:
            gx = (g2 @ W).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            rows, S = g2.shape[0], _split_count(g2.shape[0], _LinearSplitK.SPLIT_ROWS)
            if S >= 4 and rows % S == 0 and 32 <= g2.shape[1] <= 1024 and 32 <= x2.shape[1] <= 1024 \
                    and g2.is_contiguous() and x2.is_contiguous():
                part = torch.bmm(g2.view(S, rows // S, -1).transpose(1, 2), x2.view(S, rows // S, -1),
                                 out_dtype=torch.float32)
                gw = part.sum(0)[:out_f, :in_f].to(ctx.wdtype)
            else:
                gw = (g2.t() @ x2)[:out_f, :in_f]
                if not gw.is_contiguous():
                    gw = gw.contiguous()
        if ctx.has_bias and ctx.needs_input_grad[2] and gb is None:
            rows = g2.shape[0]
            if g2.shape[1] <= 8 and rows % 1024 == 0 and rows >= 8192 and g2.is_contiguous():
                # a (rows x 1) column summed to one value is a single-workgroup reduction in ATen (60 us at 65 536
                # rows); two stages keep the whole chip busy
                gb = g2.view(rows // 1024, 1024, -1).float().sum(1).sum(0)[:out_f].to(g2.dtype)
            else:
                gb = g2.sum(0)[:out_f]
        return gx, gw, gb, None, None, None


def _dense_layer_grads(g2, gbf, xin, W, out_f, in_f, wdt, need_x, need_w, need_b, rows_gemm_ws=None):
    """Gradients of y = xin @ W^T + b from g2 = dL/dy (rows, padded width; ``gbf``: its fp32 column sums when a fused
    ReLU-backward already produced them): (dL/dxin, dL/dW, dL/db), None where not needed.  ``rows_gemm_ws``: W already in
    fragment order for trs_rows_gemm (F_.rows_gemm_pack at forward time)."""
    rows = xin.shape[0]
    gw_out = gb_out = None

    def input_grad():
        if need_x and F_.rows_gemm_supported(g2, W, out_f, xin.shape[1]) and W.shape[1] == xin.shape[1]:
            # wide input, short contraction (2496 <- 400): our own kernel, K not padded to the library's tile
            return F_.rows_gemm(g2, W, out_f, xin.shape[1], packed_ws=rows_gemm_ws)
        return (g2 @ W) if need_x else None

    # GX_LAST: the input gradient is enqueued BEHIND the weight gradient, i.e. right in front of its consumer (the bucket
    # walk of the embedding backward reads the (B,N,E) gradient next): what of it is still in the Infinity Cache then
    # does not come from HBM
    gx = None if GX_LAST else input_grad()
    S = _split_count(rows, _LinearSplitK.SPLIT_ROWS)
    if need_w or (need_b and gbf is not None):
        if need_w and S >= 4 and rows % S == 0 and 32 <= g2.shape[1] <= 1024 and 32 <= xin.shape[1] <= 4096 \
                and xin.is_contiguous():
            gw = torch.empty(out_f, in_f, dtype=wdt, device=xin.device)
            St = _split_count(rows, _MLPStack.SPLIT_ROWS_WIDE)
            if xin.shape[1] >= 2 * g2.shape[1] and St >= 4:
                # wide input (the 2496-wide first layer): slices of x^T g, 16 of them at 65 536 rows -- hipBLASLt
                # runs that orientation in 190 us against 282 us for 32 slices of g^T x (tools/wgrad_probe3.py)
                part = torch.bmm(xin.view(St, rows // St, -1).transpose(1, 2), g2.view(St, rows // St, -1),
                                 out_dtype=torch.float32)
                with_b = need_b and gbf is not None
                gb = torch.empty(out_f, dtype=wdt, device=xin.device) if with_b else None
                F_.call("trs_wgrad_finish_t", F_.ptr(part), St, part.shape[1], part.shape[2], out_f, in_f,
                        F_.value_dtype_code(gw), F_.ptr(gw), F_.ptr(gbf) if with_b else F_.ptr(None), F_.ptr(gb),
                        F_.stream_ptr())
            else:
                part = torch.bmm(g2.view(S, rows // S, -1).transpose(1, 2), xin.view(S, rows // S, -1),
                                 out_dtype=torch.float32)
                with_b = need_b and gbf is not None and out_f <= (in_f + 255) // 256 * 256
                gb = torch.empty(out_f, dtype=wdt, device=xin.device) if with_b else None
                F_.call("trs_wgrad_finish", F_.ptr(part), S, part.shape[1], part.shape[2], out_f, in_f,
                        F_.value_dtype_code(gw), F_.ptr(gw), F_.ptr(gbf) if with_b else F_.ptr(None), F_.ptr(gb),
                        F_.stream_ptr())
            gw_out = gw
            if with_b:
                gb_out = gb
        elif need_w:
            gw = (g2.t() @ xin)[:out_f, :in_f]
            gw_out = gw if gw.is_contiguous() else gw.contiguous()
    if need_b and gb_out is None:
        gb_out = gbf[:out_f].to(wdt) if gbf is not None else g2.sum(0)[:out_f]
    if GX_LAST:
        gx = input_grad()
    return gx, gw_out, gb_out


class _HybridMLP(torch.autograd.Function):
    """A deep branch whose first layer is too wide for the fused kernel, as ONE autograd node: first Linear + ReLU on
    hipBLASLt, every layer behind it in trs_mlp_fused_fwd / _bwd_data (F_._FusedMLPTail's arrangement).  One node instead
    of two so that the first layer's ReLU-backward and bias gradient come out of the fused backward kernel (it masks its
    input gradient with the sign bits of h1 the forward kernel took while loading h1, and sums the columns) instead of a
    pass of their own over the (rows, 400) gradient: trs_relu_bwd_bias, 32 us of a 1.4 ms DeepFM step.
    ``tensors``: weight, bias, w_use, b_use of the first layer, then of every tail layer (w_use / b_use: the zero-padded
    copies the kernels read, or None)."""

    @staticmethod
    def forward(ctx, x, w_narrow, *tensors):
        cur = x.reshape(-1, x.shape[-1])
        w1, b1, w1u, b1u = tensors[:4]
        tail = tensors[4:]
        W1 = w1 if w1u is None else w1u
        B1 = b1 if w1u is None else b1u
        L = len(tail) // 4
        Ws = [(tail[4 * l] if tail[4 * l + 2] is None else tail[4 * l + 2]).contiguous() for l in range(L)]
        bs = [(tail[4 * l + 1] if tail[4 * l + 3] is None else tail[4 * l + 3]).contiguous() for l in range(L)]
        # TRS_HOIST_PACK (off by default, see HOIST_PACK): every copy of weights into MFMA fragment order this node needs --
        # the tail's forward and backward kernels, the first layer's input-gradient kernel -- depends on the parameters
        # only and can be ONE launch in front of (or, on the "pack" side stream, beside) the first layer's GEMM.
        rows, wpack = cur.shape[0], None
        widths_t = [W1.shape[0]] + [w.shape[0] for w in Ws]
        if HOIST_PACK and rows >= PAD_MIN_ROWS:
            fam_t = F_.mlp_fused_family(widths_t, rows)
            need_bwd = any(ctx.needs_input_grad)
            gx_ws = need_bwd and ctx.needs_input_grad[0] and W1.is_contiguous() and W1.shape[1] == cur.shape[1] and \
                F_.rows_gemm_supported_for(rows, W1.shape[0], W1, w1.shape[0], cur.shape[1])

            gemm = (W1, W1.shape[0], w1.shape[0], cur.shape[1]) if gx_ws else None
            if HOIST_PACK == 2:
                wpack, ev = F_.fused_mlp_pack_branch(Ws, bs, widths_t, rows, fam_t, need_bwd, gemm), None
            else:
                wpack, ev, side = F_.run_on_side(cur.device, "pack", lambda: F_.fused_mlp_pack_branch(
                    Ws, bs, widths_t, rows, fam_t, need_bwd, gemm))
        h1 = torch._addmm_activation(B1, cur, W1.t(), use_gelu=False)
        if wpack is not None:
            if ev is not None:
                main = F_._abi.current_stream_of(cur.device)
                main.wait_event(ev)
                for t in wpack:
                    if t is not None:
                        t.record_stream(main)      # allocated under the side stream, read (and freed) under this one
            y, hidden, masks, mask_in, fam = F_.fused_mlp_forward_raw(h1, Ws, bs, input_mask=True, family=fam_t,
                                                                      packed_ws=wpack[0])
        elif w_narrow is not None:
            # mixed family on the first h_n columns of the 512-wide rows (see _forward_hybrid)
            y, hidden, masks, mask_in, fam = F_.fused_mlp_forward_raw(h1, [w_narrow] + Ws[1:], bs, input_mask=True,
                                                                      family=F_.MLP_FAMILY_MIXED, x_stride=h1.shape[1])
        else:
            y, hidden, masks, mask_in, fam = F_.fused_mlp_forward_raw(h1, Ws, bs, input_mask=True)
        ctx.wpack = wpack
        out_f = tail[4 * (L - 1)].shape[0]
        ctx.save_for_backward(cur, W1, h1, mask_in, *Ws, *hidden, *masks)
        ctx.meta = (L, [h1.shape[1]] + [w.shape[0] for w in Ws], [tuple(tail[4 * l].shape) for l in range(L)],
                    [tail[4 * l].dtype for l in range(L)], tuple(w1.shape), w1.dtype, tuple(x.shape), fam)
        y = y[:, :out_f] if out_f != y.shape[1] else y      # a view of the padded output (the head reads it strided)
        return y.reshape(*x.shape[:-1], out_f)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        L, widths, wshapes, wdt, w1shape, w1dt, xshape, fam = ctx.meta
        saved = ctx.saved_tensors
        cur, W1, h1, mask_in = saved[:4]
        Ws = saved[4:4 + L]
        hidden, masks = saved[4 + L:3 + L + L], saved[3 + L + L:]
        needs = ctx.needs_input_grad
        rows, dev = h1.shape[0], h1.device
        gy = gy.reshape(rows, -1)
        gy2 = F_.pad_cols(gy, widths[L]) if gy.shape[1] != widths[L] else gy.contiguous()
        wpack = ctx.wpack if ctx.wpack is not None else (None, None, None)
        g1, gz, gb, gb1 = F_.fused_mlp_backward_raw(gy2, widths, Ws, masks, mask_in, family=fam, packed_ws=wpack[1])
        def layer_grads(l, splits_div=1):
            inp = h1 if l == 0 else hidden[l - 1]
            g = gy2 if l == L - 1 else gz[l]
            out_f, in_f = wshapes[l]
            return F_._tail_layer_grads(g, inp, out_f, in_f, wdt[l], gb[l], needs[6 + 4 * l], needs[7 + 4 * l], splits_div)

        def tail_grads():
            # WGRAD_PAIR: the first two tail layers' weight gradients (400 x 400 from 65 536 rows each: 256 workgroups of
            # 1024 rows, where start-up and the fp32 partial cost as much as the loop) as two launches of HALF as many
            # workgroups over twice the rows, one on the "wgrad" side stream, sharing the chip
            pair = (WGRAD_PAIR and L >= 3 and rows >= PAD_MIN_ROWS and wshapes[0] == wshapes[1]
                    and all(needs[6 + 4 * l] for l in (0, 1))
                    and F_.wgrad_rows_splits(gz[0], h1, *wshapes[0]) >= 16
                    and F_.wgrad_rows_splits(gz[1], hidden[0], *wshapes[1]) >= 16)
            if not pair:
                return [t for l in range(L) for t in (*layer_grads(l), None, None)]
            (gw1_, gb1_), ev1, side1 = F_.run_on_side(dev, "wgrad", lambda: layer_grads(1, 2))
            for t in (gz[1], hidden[0], gb[1]):
                t.record_stream(side1)
            per_layer = [layer_grads(0, 2), (gw1_, gb1_)] + [layer_grads(l) for l in range(2, L)]
            main_ = F_._abi.current_stream_of(dev)
            main_.wait_event(ev1)
            for t in (gw1_, gb1_):
                if t is not None:
                    t.record_stream(main_)
            return [t for pr in per_layer for t in (*pr, None, None)]

        # The tail's weight gradients (three GEMMs over the rows + their finish passes: six launches nobody waits for
        # until the step ends) on the "wgrad" side stream, beside the first layer's input- and weight-gradient GEMMs;
        # this node's stream waits for them before it returns (the gradients go to AccumulateGrad on this stream).
        ev = None
        if WGRAD_STREAM and rows >= PAD_MIN_ROWS:
            grads, ev, side = F_.run_on_side(dev, "wgrad", tail_grads)
        else:
            grads = tail_grads()
        gx, gw1, gbias1 = _dense_layer_grads(g1, gb1, cur, W1, w1shape[0], w1shape[1], w1dt, needs[0], needs[2], needs[3],
                                             rows_gemm_ws=wpack[2])
        if ev is not None:
            main = F_._abi.current_stream_of(dev)
            main.wait_event(ev)
            for t in grads:
                if t is not None:
                    t.record_stream(main)
        return (gx.reshape(xshape) if needs[0] else None, None, gw1, gbias1, None, None, *grads)


class _MLPStack(torch.autograd.Function):
    """A whole bf16 Linear+ReLU stack (hidden layers Linear -> ReLU, then an output Linear) as ONE autograd node: the
    same GEMM arrangement as ``_LinearSplitK`` (bias+ReLU epilogue, zero-padded widths, split-K weight gradient,
    fused ReLU-backward + bias gradient, row-dot logit layer), but one Python forward / backward instead of one per
    layer and one HIP pass (trs_wgrad_finish) for the sum + slice + cast of each split-K weight gradient.  At batch
    65 536 the DeepFM step is bound by the host's enqueue rate, not by the GPU, so launches and Python frames saved
    here are step time saved.
    ``spec``: per layer (fuse_relu, rowdot); ``tensors``: per layer weight, bias, w_use, b_use (the last two: the
    zero-padded copies the GEMMs read, or None)."""

    # rows per split-K slice of a wide-input layer's weight gradient (TRS_SPLIT_ROWS_WIDE: tuning)
    SPLIT_ROWS_WIDE = int(os.environ.get("TRS_SPLIT_ROWS_WIDE", "4096"))

    @staticmethod
    def forward(ctx, x, spec, *tensors):
        cur = x.reshape(-1, x.shape[-1])
        saved, wmeta = [], []
        for l, (fuse, rowdot) in enumerate(spec):
            weight, bias, w_use, b_use = tensors[4 * l:4 * l + 4]
            W = weight if w_use is None else w_use
            Bv = bias if w_use is None else b_use
            if rowdot:
                Wf = W.reshape(-1)
                out = torch.empty(cur.shape[0], 1, dtype=cur.dtype, device=cur.device)
                F_.call("trs_rowdot_fwd", F_.ptr(cur), F_.ptr(Wf), F_.ptr(Bv), cur.shape[0], cur.shape[1],
                        F_.value_dtype_code(cur), F_.ptr(out), F_.stream_ptr())
                W = Wf
            elif fuse:
                out = torch._addmm_activation(Bv, cur, W.t(), use_gelu=False)
            else:
                out = torch.addmm(Bv, cur, W.t())
            saved += [cur, W, out if fuse else None]
            wmeta.append((tuple(weight.shape), weight.dtype))
            cur = out
        ctx.spec, ctx.wmeta, ctx.xshape = spec, wmeta, tuple(x.shape)
        ctx.save_for_backward(*saved)
        return cur.reshape(*x.shape[:-1], cur.shape[-1])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        saved, spec = ctx.saved_tensors, ctx.spec
        needs = ctx.needs_input_grad
        grads = [None] * (4 * len(spec))
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        for l in range(len(spec) - 1, -1, -1):
            xin, W, y = saved[3 * l:3 * l + 3]
            fuse, rowdot = spec[l]
            (out_f, in_f), wdt = ctx.wmeta[l]
            need_x = l > 0 or needs[0]
            need_w, need_b = needs[2 + 4 * l], needs[3 + 4 * l]
            rows = xin.shape[0]
            if rowdot:
                C = xin.shape[1]
                gh = torch.empty_like(xin) if need_x else None
                gwf = ws = None
                ws_bytes = 0
                if need_w or need_b:
                    gwf = torch.empty(C + 1, dtype=torch.float32, device=xin.device)
                    ws_bytes = F_.size_query("trs_rowdot_bwd_workspace_bytes", rows, C)
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xin.device)
                F_.call("trs_rowdot_bwd", F_.ptr(g2), F_.ptr(xin), F_.ptr(W), rows, C, F_.value_dtype_code(xin),
                        F_.ptr(gh), F_.ptr(gwf), F_.ptr(gwf[C:]) if gwf is not None else F_.ptr(None), F_.ptr(ws),
                        ws_bytes, F_.stream_ptr())
                if need_w or need_b:
                    gq = gwf.to(wdt)                      # one cast for weight row and bias
                    if need_w:
                        grads[4 * l] = gq[:in_f].reshape(out_f, in_f)
                    if need_b:
                        grads[4 * l + 1] = gq[C:]
                g2 = gh
                continue
            gbf = None
            if fuse:
                g2, gbf = F_.relu_bwd_bias(g2, y)
            gx, grads[4 * l], grads[4 * l + 1] = _dense_layer_grads(g2, gbf, xin, W, out_f, in_f, wdt, need_x, need_w,
                                                                    need_b)
            g2 = gx
        return (g2.reshape(ctx.xshape) if needs[0] else None, None, *grads)


_strided_ok = [0]


class strided_outputs:
    """Inside this context a ``MultilayerPerceptionLayer`` may return a strided VIEW of its padded output (an (B,1)
    logit column with row stride 8): the consumer is ``functional.ctr_logit``, which reads it where it lies.  Everywhere
    else the layer returns a contiguous tensor like the reference's DNNLayer (``.view`` works, the padded buffer is
    released)."""

    def __enter__(self):
        _strided_ok[0] += 1
        return self

    def __exit__(self, *exc):
        _strided_ok[0] -= 1
        return False


def _public_out(t: torch.Tensor) -> torch.Tensor:
    if _strided_ok[0] or t.is_contiguous():
        return t
    names = t.names if t.has_names() else None
    c = (t.rename(None) if names else t).contiguous()
    if names:
        c.names = names
    return c


class MultilayerPerceptionLayer(BaseLayer):
    """Linear/activation/dropout stack + output Linear.  layers/ctr/multilayer_perceptron.py:24-84.
    Plain GEMMs: stays on nn.Linear parameters and hipBLASLt kernels (outside the hand-written path, inside the timed
    step); on a HIP device with bf16/fp16 parameters the weight gradients use a split-K batched GEMM."""

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

    def _forward_stacked(self, outputs: torch.Tensor, mods) -> Optional[torch.Tensor]:
        """The whole stack through one autograd node (_MLPStack) when it is Linear -> ReLU ... -> Linear with biases
        (inactive Dropout modules are skipped); None when the stack has any other shape."""
        lin = [m for m in mods if not isinstance(m, nn.Dropout)]
        spec, tensors, padded, mods_used = [], [], [], []
        width = outputs.shape[-1]
        i = 0
        while i < len(lin):
            mod = lin[i]
            if not isinstance(mod, nn.Linear) or mod.bias is None or mod.weight.dtype != outputs.dtype:
                return None
            fuse = i + 1 < len(lin) and type(lin[i + 1]) is nn.ReLU
            last = i + (2 if fuse else 1) >= len(lin)
            if not fuse and not last:
                return None
            out_pad = _pad_width(mod.out_features) if not last else mod.out_features
            row_bytes = out_pad * outputs.element_size()
            if fuse and (row_bytes % 16 != 0 or row_bytes > 4096):       # trs_relu_bwd_bias row limits
                return None
            padded.append((mod, width, out_pad) if (width != mod.in_features or out_pad != mod.out_features) else None)
            rowdot = (last and not fuse and mod.out_features == 1
                      and (width * outputs.element_size()) % 16 == 0
                      and F_.rowdot_width_supported(width, outputs.element_size()))
            spec.append((fuse, rowdot))
            mods_used.append(mod)
            width = out_pad
            i += 2 if fuse else 1
        if not spec or not outputs.is_contiguous():
            return None
        copies = iter(_PaddedLinear.get_many([p for p in padded if p is not None]))     # one launch for the whole stack
        for mod, p in zip(mods_used, padded):
            w_use, b_use = next(copies) if p is not None else (None, None)
            tensors += [mod.weight, mod.bias, w_use, b_use]
        out = _MLPStack.apply(outputs, tuple(spec), *tensors)
        if out.dim() == 2:
            out.names = ('B', 'O',)
        elif out.dim() == 3:
            out.names = ('B', 'N', 'O',)
        return out

    def _forward_hybrid(self, outputs: torch.Tensor, mods) -> Optional[torch.Tensor]:
        """A first layer too wide for the fused kernel (the 2496-wide input of the DeepFM / xDeepFM deep branch) on
        hipBLASLt, every layer behind it -- ReLU hidden layers and the output Linear -- as ONE kernel per direction
        (F_._FusedMLPTail): 65 536 x 400 x 400 GEMMs are too small for the library's 256-wide macro tiles (300-450
        TFLOP/s each, plus a ReLU-backward pass per layer).  None when the stack does not have that shape."""
        lin = [m for m in mods if not isinstance(m, nn.Dropout)]
        layers = []
        i = 0
        while i < len(lin):
            mod = lin[i]
            if not isinstance(mod, nn.Linear) or mod.bias is None or mod.weight.dtype != outputs.dtype:
                return None
            relu = i + 1 < len(lin) and type(lin[i + 1]) is nn.ReLU
            last = i + (2 if relu else 1) >= len(lin)
            if relu == last:
                return None
            layers.append(mod)
            i += 2 if relu else 1
        if len(layers) < 3 or not HYBRID_MLP or not outputs.is_contiguous() or outputs.dim() != 2:
            return None
        first, tail = layers[0], layers[1:]
        h_pad = _pad_width(first.out_features)
        if (h_pad * outputs.element_size()) % 16 != 0 or h_pad * outputs.element_size() > 4096:
            return None
        out_pad = max(8, (tail[-1].out_features + 7) // 8 * 8)
        widths = [h_pad] + [m.out_features for m in tail[:-1]] + [out_pad]
        if any(a.in_features != b.out_features for a, b in zip(tail, layers[:-1])):
            return None
        if first.in_features <= 512 or not F_.mlp_fused_supported_for(outputs.shape[0], outputs.dtype, outputs.is_cuda,
                                                                       widths):
            return None
        padded = [(first, outputs.shape[-1], h_pad) if (outputs.shape[-1] != first.in_features
                                                         or h_pad != first.out_features) else None,
                  (tail[0], h_pad, tail[0].out_features) if h_pad != tail[0].in_features else None]
        if len(tail) > 1:
            padded += [None] * (len(tail) - 2)
            padded.append((tail[-1], tail[-1].in_features, out_pad) if out_pad != tail[-1].out_features else None)
        elif out_pad != tail[0].out_features:
            padded[1] = (tail[0], h_pad, out_pad)
        # The library GEMMs of the first layer want its output 512 wide (PAD_MULTIPLE), the fused tail wants 416 (13 chunks
        # of 32: 19 % fewer MFMAs in its first layer, and the shape the row-owner forward is built for).  Both: the tail's
        # FORWARD reads the first 416 columns of the 512-wide rows (x_stride) against a 416-wide copy of its first weight
        # when the library resolves that stack to the mixed family; the backward keeps the 512-wide form, whose extra
        # columns meet zero weights (the gradient the first layer's weight-gradient GEMM reads stays 512 wide).
        narrow = None
        h_n = (first.out_features + 31) // 32 * 32
        if (MIXED_TAIL and len(tail) >= 2 and h_n < h_pad and len(tail) + 1 <= 8 and HYBRID_ONE_NODE
                and F_.mlp_fused_family([h_n] + widths[1:], outputs.shape[0]) == F_.MLP_FAMILY_MIXED):
            narrow = (tail[0], h_n, tail[0].out_features)
        copies = iter(_PaddedLinear.get_many([p for p in padded if p is not None] + ([narrow] if narrow else [])))
        use = [next(copies) if p is not None else (None, None) for p in padded]
        w_narrow = next(copies)[0] if narrow else None
        tensors = [first.weight, first.bias, use[0][0], use[0][1]]
        for mod, (w_use, b_use) in zip(tail, use[1:]):
            tensors += [mod.weight, mod.bias, w_use, b_use]
        if len(tail) + 1 <= 8 and HYBRID_ONE_NODE:
            out = _HybridMLP.apply(outputs, w_narrow, *tensors)
        else:
            h1 = _MLPStack.apply(outputs, ((True, False),), *tensors[:4])
            out = F_._FusedMLPTail.apply(h1, *tensors[4:])
        out.names = ('B', 'O',)
        return out

    def _forward_fused(self, outputs: torch.Tensor, mods) -> Optional[torch.Tensor]:
        """Narrow stacks (every width <= 512: the per-field MLP of DeepAndCrossNetwork, 64 -> 400 -> 400 -> 400 -> 64 on
        B*N rows) as ONE kernel per direction with the activations of a row tile kept in LDS (trs_mlp_fused_*,
        SURVEY.md 8f N4); None when the stack is not Linear -> ReLU ... -> Linear with biases or does not fit."""
        lin = [m for m in mods if not isinstance(m, nn.Dropout)]
        Ws, bs = [], []
        i = 0
        while i < len(lin):
            mod = lin[i]
            if not isinstance(mod, nn.Linear) or mod.bias is None or mod.weight.dtype != outputs.dtype:
                return None
            relu = i + 1 < len(lin) and type(lin[i + 1]) is nn.ReLU
            last = i + (2 if relu else 1) >= len(lin)
            if relu == last:                      # hidden layers carry a ReLU, the output layer does not
                return None
            Ws.append(mod.weight)
            bs.append(mod.bias)
            i += 2 if relu else 1
        if len(Ws) < 2:
            return None
        widths = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
        if any(a.shape[1] != b for a, b in zip(Ws[1:], widths[1:-1])) or outputs.shape[-1] != widths[0]:
            return None
        if not F_.mlp_fused_supported(outputs, widths):
            return None
        out = F_.fused_mlp(outputs, Ws, bs)
        if out.dim() == 2:
            out.names = ('B', 'O',)
        elif out.dim() == 3:
            out.names = ('B', 'N', 'O',)
        return out

    def forward(self, emb_inputs: torch.Tensor) -> torch.Tensor:
        outputs = _strip(emb_inputs)
        split_k = outputs.is_cuda and outputs.dtype in (torch.bfloat16, torch.float16) and torch.is_grad_enabled()
        mods = list(self.model)
        rows = outputs.numel() // max(1, outputs.shape[-1])
        # zero-padded hidden widths: only for plain Linear(+ReLU) stacks (an active Dropout would draw a different
        # random stream on the wider activations)
        pad = split_k and rows >= PAD_MIN_ROWS and all(
            isinstance(m, (nn.Linear, nn.ReLU)) or (isinstance(m, nn.Dropout) and (m.p == 0.0 or not m.training))
            for m in mods)
        width = outputs.shape[-1]          # current (possibly padded) activation width
        if pad and outputs.dtype == torch.bfloat16:
            fused = self._forward_fused(outputs, mods)
            if fused is not None:
                return _public_out(fused)
            hybrid = self._forward_hybrid(outputs, mods)
            if hybrid is not None:
                return _public_out(hybrid)
            stacked = self._forward_stacked(outputs, mods)
            if stacked is not None:
                return _public_out(stacked)
        i = 0
        while i < len(mods):
            mod = mods[i]
            if split_k and isinstance(mod, nn.Linear):
                # Linear followed by a plain nn.ReLU: the activation and its backward are folded into the Function
                fuse = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                last = not any(isinstance(m, nn.Linear) for m in mods[i + 1:])
                out_pad = _pad_width(mod.out_features) if pad and not last else mod.out_features
                if pad and (width != mod.in_features or out_pad != mod.out_features):
                    w_pad, b_pad = _PaddedLinear.get(mod, width, out_pad)
                else:
                    w_pad = b_pad = None
                if (mod.out_features == 1 and not fuse and mod.bias is not None
                        and F_.rowdot_supported(outputs.reshape(-1, outputs.shape[-1]))):
                    # the logit layer: a row-wise dot product, not a GEMM (trs_rowdot_*)
                    outputs = F_._RowDot.apply(outputs, mod.weight, mod.bias, w_pad, b_pad)
                else:
                    outputs = _LinearSplitK.apply(outputs, mod.weight, mod.bias, fuse, w_pad, b_pad)
                width = out_pad
                i += 2 if fuse else 1
            else:
                outputs = mod(outputs)
                i += 1
        if outputs.dim() == 2:
            outputs.names = ('B', 'O',)
        elif outputs.dim() == 3:
            outputs.names = ('B', 'N', 'O',)
        return outputs


# aliases, layers/ctr/__init__.py:23-35
AFMLayer = AttentionalFactorizationMachineLayer
FMLayer = FactorizationMachineLayer
FFMLayer = FieldAwareFactorizationMachineLayer
CINLayer = CompressInteractionNetworkLayer
DenseLayer = MultilayerPerceptionLayer
DNNLayer = MultilayerPerceptionLayer
FullyConnectLayer = MultilayerPerceptionLayer
FeedForwardLayer = MultilayerPerceptionLayer
