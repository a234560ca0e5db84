"""Host-side logic that needs no GPU: constructor signatures, state_dict keys (vs the reference's, captured
in tests/golden/keys.npz), offsets, lengths, aliases, error behaviour, patch()."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

from oracle import cpu_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        assert lay.DNNLayer is L.MultilayerPerceptionLayer      # the models' deep branch
        assert inp.MultiIndicesEmbedding is torecsys_amd.inputs.MultiIndicesEmbedding
        torecsys_amd.unpatch()
        assert lay.FMLayer is Old and mdl.FMLayer is Old and inp.MultiIndicesEmbedding is Old and lay.DNNLayer is Old
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


REFERENCE = "/root/reference"


def _import_real_reference():
    """The stub recipe of SURVEY 8c (tests/golden/make_golden.py): skip torecsys/__init__.py, stub torchvision."""
    import importlib
    saved = {k: sys.modules.get(k) for k in ("torecsys", "torchvision", "torchvision.transforms")}
    pkg = types.ModuleType("torecsys")
    pkg.__path__ = [REFERENCE + "/torecsys"]
    sys.modules["torecsys"] = pkg
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    mods = [importlib.import_module("torecsys." + m) for m in ("inputs", "layers", "models")]
    return pkg, mods, saved


@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE + "/torecsys"),
                    reason="the reference never travels to the GPU box: build-container test")
def test_patch_on_the_real_reference_models():
    """patch() on the REAL reference: the four north-star models (models/ctr/factorization_machine.py:35,
    deep_fm.py:46-53, deep_and_cross_network.py:43-56, xdeep_fm.py:60-79) and the Inputs router build on torecsys_amd
    classes for every hot-path child -- interaction layer, deep MLP, embeddings, router -- with the reference's
    state_dict keys, and a patched MultiIndicesEmbedding fuses the FM term by default."""
    import warnings
    import torecsys_amd
    from torecsys_amd import inputs as I, layers as L
    warnings.filterwarnings("ignore")
    pkg, (ref_inputs, ref_layers, ref_models), saved = _import_real_reference()
    try:
        N, E = 5, 8
        sizes = [7, 3, 11, 5, 9]

        def build_all():
            mods = {
                "fm": ref_models.FactorizationMachineModel(use_bias=True, dropout_p=0.0),
                "deepfm": ref_models.DeepFactorizationMachineModel(embed_size=E, num_fields=N, deep_layer_sizes=[16, 16],
                                                                   fm_dropout_p=0.0),
                "dcn": ref_models.DeepAndCrossNetworkModel(inputs_size=E, num_fields=N, deep_output_size=4,
                                                           deep_layer_sizes=[16, 16], cross_num_layers=3),
                "xdeepfm": ref_models.XDeepFactorizationMachineModel(embed_size=E, num_fields=N, cin_layer_sizes=[6, 6],
                                                                     deep_layer_sizes=[16, 16]),
            }
            emb = ref_inputs.MultiIndicesEmbedding(embed_size=E, field_sizes=sizes)
            feat = ref_inputs.MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
            emb.set_schema(["c%d" % i for i in range(N)])
            feat.set_schema(["c%d" % i for i in range(N)])
            router = ref_inputs.Inputs(schema={"feat_inputs": feat, "emb_inputs": emb})
            return mods, emb, router

        ref_mods, ref_emb, ref_router = build_all()                       # the reference's own classes
        ref_keys = {k: list(m.state_dict().keys()) for k, m in ref_mods.items()}
        ref_router_keys = list(ref_router.state_dict().keys())
        assert type(ref_mods["deepfm"].deep).__module__.startswith("torecsys.")

        torecsys_amd.patch(pkg)
        try:
            mods, emb, router = build_all()
            assert isinstance(mods["fm"].fm, L.FactorizationMachineLayer)
            assert isinstance(mods["deepfm"].fm, L.FactorizationMachineLayer)
            assert isinstance(mods["deepfm"].deep, L.MultilayerPerceptionLayer)
            assert isinstance(mods["dcn"].cross, L.CrossNetworkLayer)
            assert isinstance(mods["dcn"].deep, L.MultilayerPerceptionLayer)       # the per-field MLP
            assert isinstance(mods["xdeepfm"].cin, L.CompressInteractionNetworkLayer)
            assert isinstance(mods["xdeepfm"].deep, L.MultilayerPerceptionLayer)
            assert isinstance(emb, I.MultiIndicesEmbedding) and emb.fuse_fm is True
            assert isinstance(router, I.Inputs)
            for alias in ("DNNLayer", "DenseLayer", "FullyConnectLayer", "FeedForwardLayer", "MultilayerPerceptionLayer"):
                assert getattr(ref_layers, alias) is L.MultilayerPerceptionLayer, alias
            for k, m in mods.items():
                assert list(m.state_dict().keys()) == ref_keys[k], k
                for c in m.modules():      # nothing of the reference's layer package is left inside the models
                    assert not type(c).__module__.startswith("torecsys.layers"), (k, type(c))
            assert list(router.state_dict().keys()) == ref_router_keys
            # reference checkpoints load unchanged
            for k, m in mods.items():
                m.load_state_dict(ref_mods[k].state_dict())
            # opt-outs
            torecsys_amd.unpatch()
            assert I.DEFAULT_FUSE_FM is False
            torecsys_amd.patch(pkg, fuse_fm=False, mlp=False, router=False)
            m2, e2, r2 = build_all()
            assert e2.fuse_fm is False and not isinstance(r2, I.Inputs)
            assert not isinstance(m2["deepfm"].deep, L.MultilayerPerceptionLayer)
            assert isinstance(m2["deepfm"].fm, L.FactorizationMachineLayer)
        finally:
            torecsys_amd.unpatch()
        m3, e3, _ = build_all()
        assert type(m3["deepfm"].fm).__module__.startswith("torecsys.layers") and not isinstance(e3, I.MultiIndicesEmbedding)
    finally:
        for k in [k for k in sys.modules if k == "torecsys" or k.startswith("torecsys.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)


def _install_cpu_standins(monkeypatch):
    """torch-CPU statements of the functional entry points the four models' drop-in modules reach, backed by the oracle
    (TEST ONLY: the product entry points are HIP kernels and refuse CPU tensors).  They keep the product's calling
    conventions -- return values, the (B,E) FM side output, the (B,C,E) contraction layout -- so that everything ABOVE
    them (the drop-in modules' forwards, the named-tensor plumbing of the reference's own model code) runs unchanged."""
    from oracle import cpu_ref as O
    from torecsys_amd import functional as F_
    calls = {"embed_fm": 0, "fm_layer": 0, "cross_network": 0, "cin_contract": 0, "gather_rows": 0}

    def _idx(idx):
        idx = idx.rename(None) if idx.has_names() else idx
        return idx.unsqueeze(-1) if idx.dim() == 1 else idx

    def embed_fm(weight, idx, offsets=None, first_weight=None, want_emb=True, opt=None, padding_idx=None):
        calls["embed_fm"] += 1
        emb = O.multi_indices_embedding(weight, _idx(idx), offsets)
        first = None if first_weight is None else O.multi_indices_embedding(first_weight, _idx(idx), offsets).sum(dim=1)
        return (emb if want_emb else None), O.fm_layer(emb), first

    def gather_rows(weight, idx, offsets=None, padding_idx=None, opt=None):
        calls["gather_rows"] += 1
        if offsets is None:
            return O.single_index_embedding(weight, _idx(idx), padding_idx)
        return O.multi_indices_embedding(weight, _idx(idx), offsets)

    def fm_layer(x):
        calls["fm_layer"] += 1
        return O.fm_layer(x)

    def cross_network(x, W, b, detach_first=True):
        calls["cross_network"] += 1
        return O.cross_network(x, list(W), list(b), detach_first)

    def cin_contract(x0, xk, Wc, bias):
        calls["cin_contract"] += 1
        return O.cin_contraction(x0.transpose(1, 2), xk.transpose(1, 2), Wc.unsqueeze(-1), bias)

    for name, fn in [("embed_fm", embed_fm), ("gather_rows", gather_rows), ("fm_layer", fm_layer),
                     ("cross_network", cross_network), ("cin_contract", cin_contract),
                     ("prefetch_row_buckets", lambda *a, **k: None)]:
        monkeypatch.setattr(F_, name, fn)
    return calls


@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE + "/torecsys"),
                    reason="the reference never travels to the GPU box: build-container test")
def test_reference_model_forwards_run_over_the_dropins(monkeypatch):
    """The REAL reference's model code drives the drop-in modules: ``patch()`` the imported reference, build its four
    north-star models and its Inputs router from ITS classes, and run forward + backward through
    ``Inputs -> model.forward`` -- the reference's own named-tensor plumbing (models/ctr/deep_fm.py:68-108
    ``flatten(('N','E'),'E')`` / ``cat(dim='O')`` / ``sum(dim='O')``, deep_and_cross_network.py:71-98, xdeep_fm.py:100-122,
    factorization_machine.py:56-71) on tensors that come out of this package's modules.  The HIP entry points are
    replaced by oracle-backed CPU stand-ins INSIDE this test only (as tests/test_dist_gloo.py does for the exchange).
    Checked: the ``_trs_fused_fm`` side channel survives the models' in-place ``names = ...`` assignments and is consumed by
    the patched FMLayer (no second FM pass); outputs are (B,1) and un-named like the reference's; logits and the
    embedding-table gradients equal the un-patched reference with the same parameters."""
    import warnings
    import torecsys_amd
    from torecsys_amd import inputs as I
    warnings.filterwarnings("ignore")
    pkg, (ref_inputs, ref_layers, ref_models), saved = _import_real_reference()
    try:
        B, N, E = 6, 5, 8
        sizes = [7, 3, 11, 5, 9]
        g = torch.Generator().manual_seed(3)
        cols = {"c%d" % i: torch.randint(0, sizes[i], (B,), generator=g) for i in range(N)}

        def build_all():
            torch.manual_seed(5)
            mods = {
                "fm": ref_models.FactorizationMachineModel(use_bias=True, dropout_p=0.0),
                "deepfm": ref_models.DeepFactorizationMachineModel(embed_size=E, num_fields=N, deep_layer_sizes=[16, 16],
                                                                   fm_dropout_p=0.0, deep_dropout_p=[0.0, 0.0]),
                "dcn": ref_models.DeepAndCrossNetworkModel(inputs_size=E, num_fields=N, deep_output_size=4,
                                                           deep_layer_sizes=[16, 16], cross_num_layers=3,
                                                           deep_dropout_p=[0.0, 0.0]),
                "xdeepfm": ref_models.XDeepFactorizationMachineModel(embed_size=E, num_fields=N, cin_layer_sizes=[6, 6],
                                                                     deep_layer_sizes=[16, 16], deep_dropout_p=[0.0, 0.0]),
            }
            emb = ref_inputs.MultiIndicesEmbedding(embed_size=E, field_sizes=sizes)
            feat = ref_inputs.MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
            emb.set_schema(["c%d" % i for i in range(N)])
            feat.set_schema(["c%d" % i for i in range(N)])
            router = ref_inputs.Inputs(schema={"feat_inputs": feat, "emb_inputs": emb})
            return mods, router

        def run(mods, router):
            out = {}
            for k, m in mods.items():
                for p in list(router.parameters()) + list(m.parameters()):
                    p.grad = None
                d = router({c: v.clone() for c, v in cols.items()})
                y = m(emb_inputs=d["emb_inputs"]) if k == "dcn" else m(**d)
                y.rename(None).sum().backward()
                grads = {n: p.grad.clone() for n, p in router.named_parameters() if p.grad is not None}
                out[k] = (y, grads)
            return out

        ref_mods, ref_router = build_all()                 # un-patched: the reference's own layers end to end
        ref_out = run(ref_mods, ref_router)

        torecsys_amd.patch(pkg)
        try:
            calls = _install_cpu_standins(monkeypatch)
            mods, router = build_all()
            assert isinstance(router, I.Inputs)
            router.load_state_dict(ref_router.state_dict())
            for k in mods:
                mods[k].load_state_dict(ref_mods[k].state_dict())
            got = run(mods, router)
            for k in mods:
                y, grads = got[k]
                y0, grads0 = ref_out[k]
                assert tuple(y.shape) == (B, 1) == tuple(y0.shape), k
                assert y.names == y0.names, (k, y.names, y0.names)
                err = float((y.rename(None) - y0.rename(None)).abs().max() / y0.rename(None).abs().max())
                assert err <= 1e-5, (k, err)
                assert set(grads) == set(grads0), (k, sorted(grads), sorted(grads0))
                for n in grads:
                    gerr = float((grads[n] - grads0[n]).abs().max() / grads0[n].abs().max().clamp_min(1e-30))
                    assert gerr <= 1e-5, (k, n, gerr)
            # the fused FM term travelled from the lookup to FMLayer through the models' own renaming: one fused lookup per
            # wide-table pass, and NOT ONE separate FM pass (fm and deepfm each have an FMLayer)
            assert calls["embed_fm"] == 4 and calls["gather_rows"] == 4 and calls["fm_layer"] == 0, calls
            assert calls["cross_network"] == 1 and calls["cin_contract"] == 2, calls
            # and without the side channel (fuse_fm off) the same models fall back to the FM layer's own pass
            torecsys_amd.unpatch()
            torecsys_amd.patch(pkg, fuse_fm=False)
            calls2 = _install_cpu_standins(monkeypatch)
            mods2, router2 = build_all()
            router2.load_state_dict(ref_router.state_dict())
            for k in mods2:
                mods2[k].load_state_dict(ref_mods[k].state_dict())
            got2 = run(mods2, router2)
            for k in mods2:
                err = float((got2[k][0].rename(None) - ref_out[k][0].rename(None)).abs().max())
                assert err <= 1e-5 * float(ref_out[k][0].rename(None).abs().max()), (k, err)
            assert calls2["embed_fm"] == 0 and calls2["fm_layer"] == 2, calls2
        finally:
            torecsys_amd.unpatch()
    finally:
        for k in [k for k in sys.modules if k == "torecsys" or k.startswith("torecsys.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)


@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE + "/torecsys"),
                    reason="the reference never travels to the GPU box: build-container test")
def test_reference_stacked_input_of_single_index_embeddings_is_one_lookup(monkeypatch):
    """The REAL reference's ``StackedInput`` (inputs/base/stacked_inp.py:94-134) over patched ``SingleIndexEmbedding``s
    inside the patched ``Inputs`` router: the router recognises the schema and issues ONE table-list lookup
    (F_.gather_rows_tables; replaced here by an oracle-backed CPU stand-in, as the other entry points are) instead of N
    lookups + a cat; output names / shape / values and every table's gradient equal the un-patched reference."""
    import warnings
    import torecsys_amd
    from oracle import cpu_ref as O
    from torecsys_amd import functional as F_
    from torecsys_amd import inputs as I
    warnings.filterwarnings("ignore")
    pkg, (ref_inputs, ref_layers, ref_models), saved = _import_real_reference()
    try:
        B, E = 9, 8
        sizes = [7, 3, 11, 5]
        g = torch.Generator().manual_seed(4)
        cols = {"c%d" % i: torch.randint(0, v, (B,), generator=g) for i, v in enumerate(sizes)}

        def build():
            torch.manual_seed(2)
            children = []
            for i, v in enumerate(sizes):
                c = ref_inputs.SingleIndexEmbedding(E, v)
                c.set_schema(["c%d" % i])
                children.append(c)
            return ref_inputs.Inputs(schema={"emb_inputs": ref_inputs.StackedInput(children)})

        def run(router):
            out = router({k: v.clone() for k, v in cols.items()})["emb_inputs"]
            (out.rename(None) * torch.arange(1, E + 1).float()).sum().backward()
            return out, {n: p.grad.clone() for n, p in router.named_parameters()}

        ref_router = build()
        y0, g0 = run(ref_router)
        torecsys_amd.patch(pkg)
        try:
            calls = {"tables": 0, "rows": 0}

            def gather_rows_tables(weights, idx):
                calls["tables"] += 1
                idx = idx.rename(None) if idx.has_names() else idx
                return torch.cat([O.single_index_embedding(w, idx[:, i:i + 1], None) for i, w in enumerate(weights)], 1)

            def gather_rows(weight, idx, offsets=None, padding_idx=None, opt=None):
                calls["rows"] += 1
                return O.single_index_embedding(weight, idx.rename(None) if idx.has_names() else idx, padding_idx)

            monkeypatch.setattr(F_, "gather_rows_tables", gather_rows_tables)
            monkeypatch.setattr(F_, "gather_rows", gather_rows)
            monkeypatch.setattr(F_, "pack_columns_supported", lambda cols: False)
            monkeypatch.setattr(I, "_on_hip", lambda t: True)
            router = build()
            assert isinstance(router, I.Inputs) and router.schema["emb_inputs"].__class__.__name__ == "StackedInput"
            assert all(type(c) is I.SingleIndexEmbedding for c in router.schema["emb_inputs"].inputs)
            router.load_state_dict(ref_router.state_dict())
            y, gr = run(router)
            assert calls == {"tables": 1, "rows": 0}, calls
            assert y.names == y0.names and tuple(y.shape) == tuple(y0.shape) == (B, len(sizes), E)
            assert torch.equal(y.rename(None), y0.rename(None))
            assert set(gr) == set(g0)
            for n in gr:
                assert torch.allclose(gr[n], g0[n], rtol=1e-6, atol=0), n
            # a child with a padding row is outside the one-launch form: the StackedInput's own forward runs
            monkeypatch.setattr(I, "STACKED_ONE_LAUNCH", False)
            run(router)
            assert calls["tables"] == 1 and calls["rows"] == len(sizes), calls
        finally:
            torecsys_amd.unpatch()
    finally:
        for k in [k for k in sys.modules if k == "torecsys" or k.startswith("torecsys.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)


@pytest.mark.parametrize("N", [2, 3, 7, 16, 32, 33, 39, 40, 48])
def test_afm_pair_tiles_cover_every_pair_once_and_are_field_disjoint(N):
    """the host-built tile schedule of the AFM backward kernel (afm_packed_tiles): each of the N(N-1)/2 pairs exactly once,
    no field twice inside a tile, and no more tiles than the round-robin rounds would take"""
    import ctypes
    import numpy as np
    from torecsys_amd import _abi
    lib = _abi.load()
    buf = np.full(112 * 16, 0xffff, dtype=np.uint16)
    nt = ctypes.c_int32(0)
    rc = lib.trs_afm_pair_tiles(N, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(nt))
    assert rc == 0 and nt.value > 0
    seen = set()
    for t in range(nt.value):
        fields = set()
        for e in buf[16 * t:16 * t + 16]:
            if e == 0xffff:
                continue
            i, j = int(e) >> 8, int(e) & 0xff
            assert 0 <= i < j < N
            assert (i, j) not in seen
            seen.add((i, j))
            assert i not in fields and j not in fields
            fields.update((i, j))
    assert len(seen) == N * (N - 1) // 2
    assert np.all(buf[16 * nt.value:] == 0xffff)
    rounds = N if N % 2 else N - 1
    assert nt.value <= rounds * ((N // 2 + 15) // 16)
    if N > 32:
        assert nt.value == -(-len(seen) // 16)          # the greedy pass reaches the lower bound for these N


@pytest.mark.parametrize("N,C,E", [(5, 4, 3), (33, 8, 4), (39, 6, 2)])
def test_cin_symmetric_fold_keeps_the_first_layer_contraction(N, C, E):
    """functional.cin_fold_symmetric: with xk = x0 the oracle's contraction gives the same (B,C,E) for the folded weights
    (and their gradient w.r.t. x0, which is what the data-gradient kernel returns as dx0 + dxk), in float64"""
    from torecsys_amd import functional as F_
    from oracle import cpu_ref as O
    g = torch.Generator().manual_seed(N)
    x = torch.randn(3, E, N, generator=g, dtype=torch.float64)
    W = torch.randn(C, N * N, generator=g, dtype=torch.float64)
    Wf = F_.cin_fold_symmetric(W.float(), N).double()
    # exactness of the fold itself is a property of real arithmetic: redo it in float64 for the comparison
    W3 = W.view(C, N, N)
    Wf64 = (torch.tril(W3) + torch.triu(W3, 1).transpose(1, 2)).reshape(C, N * N)
    assert float((Wf - Wf64).abs().max()) <= 1e-6 * float(W.abs().max()) * 2
    assert float(torch.triu(Wf64.view(C, N, N), 1).abs().max()) == 0.0
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = O.cin_contraction(xa, xa, W.unsqueeze(-1), None)
    yb = O.cin_contraction(xb, xb, Wf64.unsqueeze(-1), None)
    assert float((ya - yb).abs().max()) <= 1e-10 * float(ya.abs().max())
    go = torch.randn(ya.shape, generator=g, dtype=torch.float64)
    (ya * go).sum().backward()
    (yb * go).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-10 * float(xa.grad.abs().max())


# kernel-name prefix -> (most VGPRs it may spill, why it is tolerated).  Everything else in csrc/ must not spill at all:
# scratch traffic in a hot loop is the first thing a review of these kernels looks for.
SPILL_ALLOWLIST = {
    "void trs::cin_cl_bwd_data_kernel<4, 4, 3, 8>": (2, "two address registers saved once per item, outside the k-loops"),
    "void trs::cross_mfma_bwd3_kernel<4, 6, false>": (10, "the NON-detached variant (faithful_grad=False) keeps one more "
                                                          "packed gradient tile; the default variant <4, 6, true> has none"),
    "trs::mlp_fused_fwd_kernel": (10, "kernel-invariant addresses saved at entry and re-read once per tile / layer "
                                      "(profiles/r04_kernels.md); none inside a k-loop"),
    "void trs::mlp_fused_bwd_kernel<true>": (5, "the variant that reads the row-owner sign bits: five kernel-invariant values "
                                                "stored once at entry, six single reloads per tile between the GEMM variants, "
                                                "none inside a k-loop (11 scratch instructions in 11.7 k lines of ISA)"),
    "void trs::pairw_reg_kernel<float, 1>": (13, "OPN 'vec' weight gradient: 64 accumulators per lane at the 128-register "
                                                 "cap of a 1024-thread workgroup (one pair per lane needs the 1024 threads)"),
    "void trs::pairw_reg_kernel<trs::bf16_t, 1>": (12, "as above"),
    "void trs::mlp_ro_kernel<trs::RoCfg<416, 400, 400, 8>, false, 416, 1>": (1, "one per-pass value saved at the start of a "
                                                                               "pass and re-read once at its end (the 416-wide "
                                                                               "input takes 104 of the 256 registers)"),
}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_unexpected_register_spills():
    """hipcc's own resource report over every file of csrc/ (the build's flags): no kernel that can be dispatched spills
    VGPRs, beyond a short allowlist whose entries say why.  Last round shipped never-selected template instantiations
    that spilled up to 589 registers and a hot kernel that had quietly grown scratch traffic."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spills
    bad = spills.spilling_kernels()
    unexpected = []
    for src, name, vs, ss, v, a in bad:
        lim = SPILL_ALLOWLIST.get(name)
        if lim is None or vs > lim[0]:
            unexpected.append((src, name, vs))
    assert not unexpected, f"kernels with VGPR spills: {unexpected}"


def test_bench_self_launches_for_more_than_one_gpu(monkeypatch):
    """`python bench.py --gpus N` (the driver's BENCH command form, no launcher around it) must start N ranks by itself:
    the process replaces itself with torch.distributed.run on 127.0.0.1; under a launcher (WORLD_SIZE set) or at N = 1
    it does not."""
    import importlib.util
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-tunableop"])       # module import must not touch TunableOp files
    spec = importlib.util.spec_from_file_location("trs_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.self_launch_argv(8, argv, {})
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py")) if os.path.join(ROOT, "bench.py") in cmd else cmd.index(os.path.abspath("bench.py"))
    assert cmd[i + 1:] == argv[1:]
    assert bench.self_launch_argv(8, argv, {"MASTER_PORT": "29777"})[cmd.index("--master-port") + 1] == "29777"
    assert bench.self_launch_argv(8, argv, {"WORLD_SIZE": "8", "RANK": "0"}) is None
    assert bench.self_launch_argv(1, argv, {}) is None
    # main() execs exactly that command
    seen = {}

    def fake_execv(path, args):
        seen["path"], seen["args"] = path, list(args)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--no-tunableop"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["path"] == sys.executable and seen["args"][seen["args"].index("--nproc-per-node") + 1] == "2"
    assert seen["args"][-5:] == ["--gpus", "2", "--steps", "4", "--no-tunableop"]
