"""Fused modules: same parameters / state_dict keys as the reference's module pairs, one kernel instead of two.

``FusedFieldAwareFM`` = ``MultiIndicesFieldAwareEmbedding`` (multi_indices_field_aware_emb.py:24-111) followed by
``FieldAwareFactorizationMachineLayer`` (field_aware_factorization_machine.py:50-94): the reference materialises a
(B, N*N, E) tensor between the two (12.8 GB at B=65 536, N=39, E=64, bf16); here the pair products are gathered
straight from the N tables.  ``EmbeddingFM`` = ``MultiIndicesEmbedding`` + ``FactorizationMachineLayer``
(+ the first-order ``MultiIndicesEmbedding(embed_size=1)`` sum) in one pass over the rows.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import functional as F_
from .inputs import BaseInput, MultiIndicesEmbedding, field_offsets


class FusedFieldAwareFM(BaseInput):
    """(B,N) indices -> (B, N(N-1)/2, E) named ('B','N','E'); parameters ``embeddings.{i}.weight``."""

    def __init__(self, embed_size: int, field_sizes: List[int], device: str = 'cpu', dropout_p: float = 0.0):
        super().__init__()
        self.num_fields = len(field_sizes)
        self.embeddings = nn.ModuleList([nn.Embedding(sum(field_sizes), embed_size) for _ in range(self.num_fields)])
        for embedding in self.embeddings:
            nn.init.xavier_uniform_(embedding.weight.data)
        self.register_buffer('offsets', field_offsets(field_sizes), persistent=False)
        self.dropout = nn.Dropout(dropout_p)
        self.length = embed_size
        self.to(device)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        idx = inputs.rename(None) if inputs.has_names() else inputs
        out = F_.ffm_fused([e.weight for e in self.embeddings], idx, self.offsets)
        out = self.dropout(out)
        out.names = ('B', 'N', 'E',)
        return out


class EmbeddingFM(nn.Module):
    """One kernel for ``emb = MultiIndicesEmbedding(E)(idx)``, ``FMLayer()(emb)`` and (optionally)
    ``MultiIndicesEmbedding(1)(idx).sum('N')``.  Holds the two input modules so their parameters keep the
    reference names (``emb.embedding.weight``, ``feat.embedding.weight``)."""

    def __init__(self, embed_size: int, field_sizes: List[int], first_order: bool = True, want_block: bool = True):
        super().__init__()
        self.emb = MultiIndicesEmbedding(embed_size=embed_size, field_sizes=field_sizes)
        self.feat = MultiIndicesEmbedding(embed_size=1, field_sizes=field_sizes) if first_order else None
        self.want_block = want_block

    def forward(self, inputs: torch.Tensor):
        idx = inputs.rename(None) if inputs.has_names() else inputs
        fw = None if self.feat is None else self.feat.embedding.weight
        emb, fm, first = F_.embed_fm(self.emb.embedding.weight, idx, self.emb.offsets, fw, self.want_block)
        if emb is not None:
            emb.names = ('B', 'N', 'E',)
        fm.names = ('B', 'O',)
        return emb, fm, first


class BCEWithLogitsLoss(nn.Module):
    """``nn.BCEWithLogitsLoss()`` (mean reduction, no weights) on the HIP device: ``loss(logits, labels)`` with logits in
    fp32 or bf16 -- no ``.float()`` cast needed in front -- and 0/1 (or soft) labels in fp32 / bf16; an fp32 scalar.
    Forward two launches, backward one (functional.bce_with_logits) instead of ATen's ~16.  The loss SURVEY.md 8d defines
    the fwd+bwd metric on; configurations this module does not cover raise at construction."""

    def __init__(self, weight=None, size_average=None, reduce=None, reduction: str = 'mean', pos_weight=None):
        super().__init__()
        if weight is not None or pos_weight is not None or reduction != 'mean' or size_average is not None \
                or reduce is not None:
            raise NotImplementedError("torecsys_amd.fused.BCEWithLogitsLoss covers reduction='mean' without weights; use "
                                      "torch.nn.BCEWithLogitsLoss for anything else")

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return F_.bce_with_logits(input, target)
