"""CPU oracle for the torecsys embedding-lookup + feature-interaction hot path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``torecsys_amd/`` imports
``oracle`` and the product path raises when the HIP library is missing.

It is a plain, name-free, op-for-op PyTorch-CPU restatement of the reference
path (p768lwy3/torecsys, pure Python on ATen), written from the reference's
behaviour; every function cites the reference ``file:line`` it follows.  The
reference's arithmetic lives in PyTorch ATen (``requirements.txt:19`` pins
``torch~=1.11``; this image has 2.10) -- ``nn.Embedding``, ``nn.Linear``,
``nn.Conv1d(k=1)``, ``nn.BatchNorm1d``, ``einsum``, ``sum`` -- so the oracle
uses the same ATen CPU ops, without named tensors and without nn.Module state.

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against
golden vectors produced by importing the real reference in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``): gather bit-exact,
float results to <=1e-6 relative (fp32).

All functions are pure: weights come in as tensors, gradients (where a test
needs them) come from autograd over these same functions.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# I1-I3: index -> embedding inputs
# --------------------------------------------------------------------------

def field_offsets(field_sizes: Sequence[int], through_float32: bool = False) -> torch.Tensor:
    """Per-field row offsets into the concatenated table, shape (N,), int64.

    Reference: torecsys/inputs/base/multi_indices_emb.py:54-57 and
    multi_indices_field_aware_emb.py:56-58 -- ``(0, *cumsum(field_sizes)[:-1])``.
    The reference builds it with ``torch.Tensor(...).long()``, i.e. through
    float32 (exact only below 2**24 rows); ``through_float32``
    reproduces that, the default computes in int64 (identical wherever the
    reference is exact).
    """
    sizes = torch.as_tensor(list(field_sizes), dtype=torch.int64)
    cum = torch.cumsum(sizes, 0)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), cum[:-1]])
    if through_float32:
        off = off.to(torch.float32).long()
    return off


def single_index_embedding(weight: torch.Tensor, idx: torch.Tensor,
                           padding_idx: Optional[int] = None) -> torch.Tensor:
    """(B,N) any-int idx -> (B,N,E).  single_index_emb.py:56-58
    (``inputs.long()`` then ``nn.Embedding``).  ``padding_idx`` only affects the
    gradient (row gets none), as in ``F.embedding``."""
    return F.embedding(idx.long(), weight, padding_idx=padding_idx)


def multi_indices_embedding(weight: torch.Tensor, idx: torch.Tensor, offsets: torch.Tensor,
                            flatten: bool = False) -> torch.Tensor:
    """(B,N) idx -> (B,N,E) (or (B,1,N*E) when ``flatten``).

    multi_indices_emb.py:103-112: ``inputs + offsets`` (int64 promotion) then one
    lookup in the concatenated ``sum(field_sizes) x E`` table."""
    g = idx + offsets.view(1, -1)
    out = F.embedding(g, weight)
    if flatten:
        out = out.reshape(out.shape[0], 1, -1)
    return out


def multi_indices_field_aware_embedding(weights: Sequence[torch.Tensor], idx: torch.Tensor,
                                        offsets: torch.Tensor) -> torch.Tensor:
    """(B,N) idx -> (B,N*N,E); block i (rows i*N..i*N+N-1) = table i applied to
    all N fields.  multi_indices_field_aware_emb.py:102-105 (cat of N lookups)."""
    g = idx + offsets.view(1, -1)
    return torch.cat([F.embedding(g, w) for w in weights], dim=1)


# --------------------------------------------------------------------------
# F1-F5: feature-interaction layers (dropout omitted: parity is at p=0 / eval)
# --------------------------------------------------------------------------

def fm_layer(x: torch.Tensor) -> torch.Tensor:
    """(B,N,E) -> (B,E): 0.5*((sum_n x)^2 - sum_n x^2).
    layers/ctr/factorization_machine.py:62-73 (same op order: sum, pow, pow, sum, sub, mul)."""
    squared_sum = x.sum(dim=1) ** 2
    sum_squared = (x ** 2).sum(dim=1)
    return 0.5 * (squared_sum - sum_squared)


def pair_indices(num_fields: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Upper-triangle (i<j) pairs in lexicographic order.
    inner_product_network.py:45-52; field_aware_factorization_machine.py:75-76."""
    rows, cols = [], []
    for i in range(num_fields - 1):
        for j in range(i + 1, num_fields):
            rows.append(i)
            cols.append(j)
    return torch.tensor(rows, dtype=torch.int64), torch.tensor(cols, dtype=torch.int64)


def ffm_layer(x: torch.Tensor, num_fields: int) -> torch.Tensor:
    """(B,N*N,E) -> (B,NC2,E): out[:,p(i,j)] = x[:,i*N+j] * x[:,j*N+i], i<j.
    field_aware_factorization_machine.py:69-87."""
    B, _, E = x.shape
    x4 = x.reshape(B, num_fields, num_fields, E)
    r, c = pair_indices(num_fields)
    return x4[:, r, c] * x4[:, c, r]


def inner_product_layer(x: torch.Tensor) -> torch.Tensor:
    """(B,N,E) -> (B,NC2): out[:,p] = sum_e x[:,i_p,e]*x[:,j_p,e].
    inner_product_network.py:68-74."""
    r, c = pair_indices(x.shape[1])
    return (x[:, r] * x[:, c]).sum(dim=-1)


def cross_network(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                  detach_first_input: bool = True) -> torch.Tensor:
    """(B,N,E) or (B,E) -> same shape.  x_{l+1} = x0 * (x_l W_l^T + b_l) + x0.

    cross_network.py:65-79.  The residual is ``+ x0`` (not ``+ x_l``), ``W_l`` is a
    full ExE ``nn.Linear`` and -- line 65 -- the running value starts from
    ``emb_inputs.detach()``, so no gradient flows through layer 0's linear input
    (SURVEY §9 Q2/Q4).  ``detach_first_input=False`` gives the textbook gradient.
    """
    out = x.detach() if detach_first_input else x
    for w, b in zip(weights, biases):
        out = F.linear(out, w, b)
        out = x * out
        out = out + x
    return out


def batchnorm_channels(y: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                       running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
                       training: bool, momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """nn.BatchNorm1d over a (B,C,E) tensor: statistics per channel over (B,E)."""
    return F.batch_norm(y, running_mean, running_var, weight, bias, training, momentum, eps)


def cin_contraction(x0: torch.Tensor, hidden: torch.Tensor, conv_weight: torch.Tensor,
                    conv_bias: Optional[torch.Tensor]) -> torch.Tensor:
    """One CIN layer's contraction: x0 (B,E,N), hidden (B,E,H) -> y (B,C,E).

    compress_interaction_network.py:125-137: the outer product over the field dims per
    (b,e), Z[b,(n,h),e] = x0[b,e,n]*hidden[b,e,h] (flatten order n*H+h, :125-132), then
    Conv1d(k=1) = a channel-mixing GEMM with ``conv_weight`` (C, N*H, 1) (:137)."""
    B, E, N = x0.shape
    H = hidden.shape[2]
    z = x0.unsqueeze(3) * hidden.unsqueeze(2)            # (B,E,N,H)
    z = z.reshape(B, E, N * H).permute(0, 2, 1)          # (B,N*H,E)
    return F.conv1d(z, conv_weight, conv_bias)           # (B,C,E)


def cin_glue(y: torch.Tensor, bn_weight: Optional[torch.Tensor], bn_bias: Optional[torch.Tensor],
             running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
             use_batchnorm: bool, is_direct: bool, training: bool, activation=torch.relu):
    """The rest of one CIN layer on the contraction result y (B,C,E) -> (z, direct, hidden (B,E,H')).

    compress_interaction_network.py:137-171: BatchNorm1d (batch statistics when ``training``),
    activation, and -- unless ``is_direct`` -- ``chunk(2, dim=1)`` into ``direct`` and the next
    layer's ``hidden`` (EVERY layer is split: the ``i != len-1`` guard at :151 is always true)."""
    if use_batchnorm:
        y = batchnorm_channels(y, bn_weight, bn_bias, running_mean, running_var, training)
    if activation is not None:
        y = activation(y)
    if is_direct:
        direct, hid = y, y
    else:
        direct, hid = torch.chunk(y, 2, dim=1)
    return y, direct, hid.permute(0, 2, 1)


def cin_layer(x: torch.Tensor,
              conv_weights: Sequence[torch.Tensor], conv_biases: Sequence[Optional[torch.Tensor]],
              fc_weight: torch.Tensor, fc_bias: torch.Tensor,
              bn_weights: Optional[Sequence[torch.Tensor]] = None,
              bn_biases: Optional[Sequence[torch.Tensor]] = None,
              bn_running_means: Optional[Sequence[torch.Tensor]] = None,
              bn_running_vars: Optional[Sequence[torch.Tensor]] = None,
              is_direct: bool = False, training: bool = True, activation=torch.relu,
              return_intermediates: bool = False):
    """Compress Interaction Network, (B,N,E) -> (B,O).

    compress_interaction_network.py:114-182: per layer ``cin_contraction`` then ``cin_glue``
    (see there); out = fc(sum_e cat(direct)) (:176-181).
    ``conv_weights[k]`` has shape (C_out, N*H_k, 1) like nn.Conv1d.
    Running statistics, if given, are updated in place like nn.BatchNorm1d.
    ``activation`` may be a list of callables, one per layer (tests use it to replay a mask).
    """
    x0 = x.permute(0, 2, 1)                      # (B,E,N)
    hidden = x0
    directs: List[torch.Tensor] = []
    inter = []
    for k, w in enumerate(conv_weights):
        pre_bn = cin_contraction(x0, hidden, w, conv_biases[k])
        act = activation[k] if isinstance(activation, (list, tuple)) else activation
        y, direct, hidden = cin_glue(
            pre_bn, None if bn_weights is None else bn_weights[k], None if bn_biases is None else bn_biases[k],
            None if bn_running_means is None else bn_running_means[k],
            None if bn_running_vars is None else bn_running_vars[k],
            bn_weights is not None, is_direct, training, act)
        directs.append(direct)
        inter.append((pre_bn, y))
    pooled = torch.cat(directs, dim=1).sum(dim=-1)           # (B,sum H)
    out = F.linear(pooled, fc_weight, fc_bias)
    if return_intermediates:
        return out, inter, pooled
    return out


def mlp(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
        activation=torch.relu) -> torch.Tensor:
    """DNNLayer = Linear/act stack + LinearOutput (no act on the last).
    layers/ctr/multilayer_perceptron.py:53-61,63-84.  Out of the hot path (plain
    GEMMs) but inside the timed DeepFM/DCN/xDeepFM step."""
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = F.linear(x, w, b)
        if i < n - 1 and activation is not None:
            x = activation(x)
    return x


# --------------------------------------------------------------------------
# M1-M4: the four caller models (test / bench harness compositions)
# --------------------------------------------------------------------------

def fm_model(feat: torch.Tensor, emb: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """models/ctr/factorization_machine.py:58-69: sum_n feat + sum_e FM(emb) + bias -> (B,1)."""
    out = fm_layer(emb).sum(dim=1, keepdim=True) + feat.sum(dim=1)
    if bias is not None:
        out = out + bias.view(1, 1)
    return out


def deepfm_model(feat: torch.Tensor, emb: torch.Tensor,
                 deep_w: Sequence[torch.Tensor], deep_b: Sequence[torch.Tensor]) -> torch.Tensor:
    """models/ctr/deep_fm.py:73-108: sum(cat[FM(emb), feat]) + DNN(flatten(emb)) -> (B,1)."""
    B = emb.shape[0]
    fm_out = torch.cat([fm_layer(emb), feat.reshape(B, -1)], dim=1).sum(dim=1, keepdim=True)
    deep_out = mlp(emb.reshape(B, -1), deep_w, deep_b)
    return deep_out + fm_out


def dcn_model(emb: torch.Tensor,
              cross_w: Sequence[torch.Tensor], cross_b: Sequence[torch.Tensor],
              deep_w: Sequence[torch.Tensor], deep_b: Sequence[torch.Tensor],
              fc_w: torch.Tensor, fc_b: torch.Tensor) -> torch.Tensor:
    """models/ctr/deep_and_cross_network.py:71-96: per-field cross and per-field DNN
    (DNN ``inputs_size`` is E, :44-50), cat on the last dim, flatten, fc -> (B,O)."""
    B = emb.shape[0]
    cross_out = cross_network(emb, cross_w, cross_b)
    deep_out = mlp(emb, deep_w, deep_b)
    cat = torch.cat([cross_out, deep_out], dim=2)
    return F.linear(cat.reshape(B, -1), fc_w, fc_b)


def xdeepfm_model(feat: torch.Tensor, emb: torch.Tensor, cin_kwargs: dict,
                  deep_w: Sequence[torch.Tensor], deep_b: Sequence[torch.Tensor],
                  bias: torch.Tensor) -> torch.Tensor:
    """models/ctr/xdeep_fm.py:100-122: sum_n feat + CIN(emb) + DNN(flatten(emb)) + bias -> (B,1)."""
    B = emb.shape[0]
    cin_out = cin_layer(emb, **cin_kwargs)
    deep_out = mlp(emb.reshape(B, -1), deep_w, deep_b)
    return feat.sum(dim=1) + cin_out + deep_out + bias


def bce_with_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """The loss SURVEY.md 8d defines the fwd+bwd metric on: ``nn.BCEWithLogitsLoss()`` (mean reduction) = ATen's
    binary_cross_entropy_with_logits, on fp32 logits.  The reference's trainer takes the criterion from the user
    (trainer/torecsys_pipeline.py:440-470); this is the one the benchmark fixes."""
    return F.binary_cross_entropy_with_logits(logits.float(), labels.float())


# --------------------------------------------------------------------------
# Row-sharded lookup (the build's multi-GPU design) -- single-process
# statement of what the all-to-all path must reproduce.
# --------------------------------------------------------------------------

def shard_bounds(num_rows: int, world: int) -> torch.Tensor:
    """Contiguous row ranges: rank r owns [r*ceil(V/W), min(V,(r+1)*ceil(V/W)))."""
    per = (num_rows + world - 1) // world
    return torch.tensor([min(num_rows, r * per) for r in range(world + 1)], dtype=torch.int64)


def sharded_lookup(shards: Sequence[torch.Tensor], gidx: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
    """Gather global row ids from a row-sharded table == lookup in the concatenated table."""
    return F.embedding(gidx, torch.cat(list(shards), dim=0))


# ---------------------------------------------------------------------------------------------
# the other pair-pattern layers (same (i<j) pair order as the inner product)
# ---------------------------------------------------------------------------------------------
def outer_product_layer(x: torch.Tensor, kernel: torch.Tensor, kernel_type: str = "mat") -> torch.Tensor:
    """(B,N,E) -> (B,NC2).  outer_product_network.py:94-129.
    'mat': kernel (E,NC2,E): out[b,p] = sum_h sum_e x[b,i_p,e] * K[h,p,e] * x[b,j_p,h]   (:107-121)
    'vec': kernel (1,NC2,E): out[b,p] = sum_e x[b,i_p,e] * x[b,j_p,e] * K[0,p,e]          (:123-129)
    'num': kernel (1,NC2,1): out[b,p] = K[0,p,0] * sum_e x[b,i_p,e] * x[b,j_p,e]."""
    r, c = pair_indices(x.shape[1])
    p, q = x[:, r], x[:, c]
    if kernel_type == "mat":
        kp = (p.unsqueeze(1) * kernel.unsqueeze(0)).sum(dim=-1)      # (B,E_h,NC2)
        return (kp.permute(0, 2, 1) * q).sum(dim=-1)
    if kernel_type in ("vec", "num"):
        return (p * q * kernel).sum(dim=-1)
    raise ValueError('kernel_type only allows: ["mat", "num", "vec"].')


def afm_layer(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
              score_keep: Optional[torch.Tensor] = None, keep_scale: float = 1.0):
    """(B,N,E) -> ((B,E), (B,NC2,1)).  attentional_factorization_machine.py:86-125:
    prod[b,p,:] = x_i * x_j;  score = Dropout(softmax_p(W2 relu(W1 prod + b1) + b2));  out[b,:] = sum_p score[b,p] prod[b,p,:].
    ``score_keep`` (B,NC2) 0/1 + ``keep_scale`` = 1/(1-p) restate nn.Dropout GIVEN its mask (the last module of
    ``self.attention``, :82: the returned scores and the weighted sum both see the dropped scores); None = eval / p = 0.
    The output dropout (:84, :120) is left to the caller."""
    r, c = pair_indices(x.shape[1])
    prod = x[:, r] * x[:, c]
    h = torch.relu(torch.nn.functional.linear(prod, w1, b1))
    attn = torch.softmax(torch.nn.functional.linear(h, w2, b2), dim=1)
    if score_keep is not None:
        attn = attn * (score_keep.to(attn.dtype) * keep_scale).unsqueeze(-1)
    return (prod * attn).sum(dim=1), attn


def bilinear_layer(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, bilinear_type: str = "all") -> torch.Tensor:
    """(B,N,E) -> (B,NC2,E).  bilinear_interaction.py:230-255.
    'all'  (:72-76):   out[b,p,:] = (x[b,i_p,:] @ W) * x[b,j_p,:] + bias,        W (E,E), bias (E)
    'each' (:144-149): out[b,p,:] = (x[b,i_p,:] @ W[p]) * x[b,j_p,:] + bias[p],  W (NC2,E,E), bias (NC2,E)."""
    r, c = pair_indices(x.shape[1])
    p, q = x[:, r], x[:, c]
    if bilinear_type == "all":
        return torch.matmul(p, weight) * q + bias
    if bilinear_type == "each":
        return torch.einsum("bpe,peh->bph", p, weight) * q + bias
    raise ValueError('bilinear_type only allows: ["all", "each", "interaction"].')
