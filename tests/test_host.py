"""Host-side logic that needs no GPU: constructor signatures, state_dict keys (vs the reference's, captured
in tests/golden/keys.npz), offsets, lengths, aliases, error behaviour, patch()."""
import sys
import types

import pytest
import torch
import torch.nn as nn

from oracle import cpu_ref as O


def test_state_dict_keys_match_reference(golden):
    from torecsys_amd import inputs as I, layers as L
    K = golden("keys")
    built = {
        "multi": I.MultiIndicesEmbedding(embed_size=4, field_sizes=[3, 4]),
        "single": I.SingleIndexEmbedding(embed_size=4, field_size=5),
        "fa": I.MultiIndicesFieldAwareEmbedding(embed_size=4, field_sizes=[3, 4]),
        "fm": L.FactorizationMachineLayer(),
        "ffm": L.FieldAwareFactorizationMachineLayer(num_fields=3),
        "ipn": L.InnerProductNetworkLayer(num_fields=3),
        "cross": L.CrossNetworkLayer(inputs_size=4, num_layers=2),
        "cin": L.CompressInteractionNetworkLayer(embed_size=4, num_fields=3, output_size=1, layer_sizes=[2, 2]),
    }
    for k, m in built.items():
        ref = [s for s in K("keys/" + k) if s]
        assert list(m.state_dict().keys()) == ref, k


def test_offsets_lengths_and_attrs():
    from torecsys_amd.inputs import MultiIndicesEmbedding, MultiIndicesFieldAwareEmbedding, SingleIndexEmbedding
    fs = [3, 5, 2, 7]
    m = MultiIndicesEmbedding(embed_size=8, field_sizes=fs)
    assert torch.equal(m.offsets, O.field_offsets(fs)) and m.offsets.dtype == torch.int64
    assert (m.field_size, m.embed_size, m.padding_idx, len(m)) == (17, 8, None, 8)
    assert len(MultiIndicesEmbedding(embed_size=8, field_sizes=fs, flatten=True)) == 32
    assert "offsets" not in m.state_dict()
    # offsets beyond 2**24 rows stay exact (the reference's float32 round trip does not)
    big = [20_000_001, 30_000_001, 5]
    assert MultiIndicesEmbedding.__init__.__defaults__ is not None
    assert O.field_offsets(big).tolist() == [0, 20_000_001, 50_000_002]
    assert O.field_offsets(big, through_float32=True).tolist() != [0, 20_000_001, 50_000_002]
    fa = MultiIndicesFieldAwareEmbedding(embed_size=4, field_sizes=fs)
    assert fa.num_fields == 4 and len(fa) == 4 and len(fa.embeddings) == 4
    s = SingleIndexEmbedding(embed_size=6, field_size=9, padding_idx=0)
    assert len(s) == 6 and s.embedding.padding_idx == 0
    pre = nn.Parameter(torch.randn(5, 3, names=("N", "E")))
    s2 = SingleIndexEmbedding(embed_size=None, field_size=None, nn_embedding=pre)
    assert len(s2) == 3 and tuple(s2.embedding.weight.shape) == (5, 3)
    m.set_schema("a")
    assert m.schema.inputs == ["a"]
    with pytest.raises(ValueError):
        MultiIndicesEmbedding()
    with pytest.raises(NotImplementedError):
        MultiIndicesEmbedding(embed_size=4, field_sizes=[2, 2], sparse=True)


def test_layer_surface():
    from torecsys_amd import layers as L
    assert L.FMLayer is L.FactorizationMachineLayer and L.FFMLayer is L.FieldAwareFactorizationMachineLayer
    assert L.CINLayer is L.CompressInteractionNetworkLayer and L.DNNLayer is L.MultilayerPerceptionLayer
    assert L.FactorizationMachineLayer(None).dropout.p == 0.0            # Q1: None -> 0.0
    assert L.FactorizationMachineLayer().inputs_size == {"inputs": ("B", "N", "E")}
    assert L.InnerProductNetworkLayer(4).row_idx.tolist() == [0, 0, 0, 1, 1, 2]
    cin = L.CompressInteractionNetworkLayer(embed_size=8, num_fields=5, output_size=2, layer_sizes=[4, 6, 3])
    assert [s.Conv1d.out_channels for s in cin.model] == [8, 12, 6]      # every layer is split (quirk Q5)
    assert [s.Conv1d.in_channels for s in cin.model] == [25, 20, 30] and cin.fc.in_features == 13
    cd = L.CompressInteractionNetworkLayer(embed_size=8, num_fields=5, output_size=2, layer_sizes=[4, 6], is_direct=True)
    assert [s.Conv1d.out_channels for s in cd.model] == [4, 6]
    with pytest.raises(ValueError):
        L.MultilayerPerceptionLayer(8, 1, [4, 4], dropout_p=[0.1])


def test_no_cpu_fallback():
    from torecsys_amd import layers as L
    from torecsys_amd.inputs import MultiIndicesEmbedding
    x = torch.randn(2, 3, 4)
    for lay, inp in ((L.FMLayer(), x), (L.InnerProductNetworkLayer(3), x), (L.CrossNetworkLayer(4, 1), x),
                     (L.FFMLayer(2), torch.randn(2, 4, 4)),
                     (L.CINLayer(4, 3, 1, [2]), x)):
        with pytest.raises(RuntimeError, match="no CPU path"):
            lay(inp)
    with pytest.raises(RuntimeError, match="no CPU path"):
        MultiIndicesEmbedding(embed_size=4, field_sizes=[2, 2])(torch.zeros(1, 2, dtype=torch.long))


def test_inputs_router_cpu_plumbing():
    """Inputs routes dict columns into (B,N) index blocks (inputs.py:69-87); checked with a recording stub."""
    from torecsys_amd.inputs import BaseInput, Inputs

    class Rec(BaseInput):
        def __init__(self):
            super().__init__()
            self.length = 1
            self.seen = None

        def forward(self, t):
            self.seen = t
            return t

    r = Rec()
    r.set_schema(["a", "b", "c"])
    out = Inputs({"x": r})({"a": torch.tensor([1, 2]), "b": torch.tensor([[3], [4]]), "c": torch.tensor([5, 6])})
    assert out["x"].tolist() == [[1, 3, 5], [2, 4, 6]]
    inp = Inputs(None)
    inp.add_inputs("y", Rec())
    with pytest.raises(AssertionError):
        inp.add_inputs("y", Rec())
    with pytest.raises(TypeError):
        inp.add_inputs(1, Rec())


def test_patch_rebinds_reference_aliases():
    import torecsys_amd
    from torecsys_amd import layers as L
    pkg = types.ModuleType("fake_trs")
    lay = types.ModuleType("fake_trs.layers")
    mdl = types.ModuleType("fake_trs.models")
    inp = types.ModuleType("fake_trs.inputs")

    class Old:  # stand-in for the reference classes
        pass
    lay.FMLayer = lay.FactorizationMachineLayer = lay.CINLayer = Old
    lay.DNNLayer = Old
    mdl.FMLayer = Old                       # `from torecsys.layers import FMLayer` inside a model module
    inp.MultiIndicesEmbedding = Old
    for m in (pkg, lay, mdl, inp):
        sys.modules[m.__name__] = m
    try:
        torecsys_amd.patch(pkg)
        assert lay.FMLayer is L.FMLayer and mdl.FMLayer is L.FMLayer and lay.CINLayer is L.CINLayer
        assert lay.DNNLayer is Old          # out-of-path layers are left alone
        assert inp.MultiIndicesEmbedding is torecsys_amd.inputs.MultiIndicesEmbedding
        torecsys_amd.unpatch()
        assert lay.FMLayer is Old and mdl.FMLayer is Old and inp.MultiIndicesEmbedding is Old
    finally:
        for m in (pkg, lay, mdl, inp):
            sys.modules.pop(m.__name__, None)


def test_split_count_rules():
    """split-K slice count of the MLP weight gradients (layers._split_count): divides the rows, at least 4 slices of
    at least ``slice_rows`` rows, never more than 384"""
    from torecsys_amd.layers import _split_count
    assert _split_count(65536, 2048) == 32 and _split_count(65536, 4096) == 16
    assert _split_count(16384, 2048) == 8 and _split_count(8192, 2048) == 4
    assert _split_count(4096, 2048) == 0                     # fewer than 4 slices: do not split
    for rows in (65536 * 39, 8192 * 39, 65536 * 39 // 3, 3 * 2 ** 20):
        s = _split_count(rows, 2048)
        assert 4 <= s <= 384 and rows % s == 0 and rows // s >= 2048
    assert _split_count(1000003, 2048) == 0                  # a prime row count cannot be cut evenly


def test_rowdot_width_rule():
    from torecsys_amd.functional import rowdot_width_supported
    assert rowdot_width_supported(512, 2) and rowdot_width_supported(8, 2) and rowdot_width_supported(128, 4)
    assert not rowdot_width_supported(400, 2)                # 50 vectors: not a power of two
    assert not rowdot_width_supported(1024, 2)               # 128 vectors: more than a wavefront
    assert not rowdot_width_supported(12, 2)                 # not a multiple of 16 bytes


def test_fused_optimizer_state_dict_roundtrip():
    """FusedSparse* state is keyed by the nn.Parameter, follows it across devices and survives a checkpoint
    (a resume silently reset accumulators / moments / step counts)."""
    import torch.nn as nn
    from torecsys_amd.optim import FusedSparseAdagrad, FusedSparseAdam
    p = nn.Parameter(torch.zeros(5, 4))
    ada = FusedSparseAdagrad(0.1, initial_accumulator_value=0.5)
    st = ada.state_for(p.data, p)
    assert st.shape == (5, 4) and float(st[0, 0]) == 0.5
    st.add_(1.0)
    assert ada.state_for(p.data, p) is st                       # same parameter -> same state, new .data tensor or not
    sd = ada.state_dict([("emb.embedding.weight", p)])
    assert list(sd["tables"]) == ["emb.embedding.weight"]
    p2 = nn.Parameter(torch.zeros(5, 4))
    ada2 = FusedSparseAdagrad(0.7)
    ada2.load_state_dict(sd, [("emb.embedding.weight", p2)])
    assert ada2.lr == 0.1 and torch.equal(ada2.state_for(p2.data, p2), st)
    adam = FusedSparseAdam(1e-3)
    for _ in range(3):
        adam.next_step_size(p.data, p)
    adam.state_for(p.data, p)[0].fill_(2.0)
    sd = adam.state_dict([("w", p)])
    adam2 = FusedSparseAdam(1e-3)
    adam2.load_state_dict(sd, {"w": p2}.items())
    assert abs(adam2.next_step_size(p2.data, p2) - adam.next_step_size(p.data, p)) < 1e-12      # both at step 4
    assert torch.equal(adam2.state_for(p2.data, p2)[0], adam.state_for(p.data, p)[0])
    with pytest.raises(KeyError):
        FusedSparseAdam().load_state_dict(sd, [("other", p2)])
