"""GPU parity: lookup (I1-I3), dense-gradient scatter, FM (F1) and the fused lookup+FM kernel, through the
C ABI, against the golden vectors and the CPU oracle.  Gather: bit-exact.  fp32 sums: 1e-5 relative.
bf16: 1e-2 relative against the fp32 oracle evaluated on bf16-rounded inputs."""
import pytest
import torch

from conftest import LAYER_SHAPES, rel_err
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

TOL32 = 1e-5
TOLBF = 1e-2


def _tag(s):
    return "%d_%d_%d" % s


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs a HIP device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_multi_indices_embedding_golden(golden, dev, shape):
    from torecsys_amd.inputs import MultiIndicesEmbedding
    G = golden("inputs")
    t = "multi/" + _tag(shape)
    fs = G(t + "/field_sizes").tolist()
    m = MultiIndicesEmbedding(embed_size=shape[2], field_sizes=fs).to(dev)
    assert list(m.state_dict().keys()) == ["embedding.weight"]
    m.embedding.weight.data.copy_(G(t + "/weight"))
    idx = G(t + "/idx").to(dev)
    out = m(idx)
    assert out.names == ("B", "N", "E")
    assert torch.equal(out.rename(None).cpu(), G(t + "/out"))          # bit-exact
    (out.rename(None) * G(t + "/gout").to(dev)).sum().backward()
    assert rel_err(m.embedding.weight.grad.cpu(), G(t + "/gweight")) <= TOL32
    # int32 indices take the same path
    out32 = m(idx.to(torch.int32))
    assert torch.equal(out32.rename(None).cpu(), G(t + "/out"))
    # flatten
    mf = MultiIndicesEmbedding(embed_size=shape[2], field_sizes=fs, flatten=True).to(dev)
    yf = mf(idx)
    assert list(yf.shape) == G(t + "/flatten_shape").tolist() and yf.names == ("B", "N", "E")
    assert [len(m), len(mf)] == G(t + "/length").tolist()
    # E = 1 first-order table (element path)
    m1 = MultiIndicesEmbedding(embed_size=1, field_sizes=fs).to(dev)
    m1.embedding.weight.data.copy_(G("multi1/" + _tag(shape) + "/weight"))
    assert torch.equal(m1(idx).rename(None).cpu(), G("multi1/" + _tag(shape) + "/out"))


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_single_index_embedding_golden(golden, dev, shape):
    from torecsys_amd.inputs import SingleIndexEmbedding
    G = golden("inputs")
    t = "single/" + _tag(shape)
    w = G(t + "/weight")
    m = SingleIndexEmbedding(embed_size=shape[2], field_size=w.shape[0], padding_idx=0).to(dev)
    assert list(m.state_dict().keys()) == ["embedding.weight"]
    m.embedding.weight.data.copy_(w)
    out = m(G(t + "/idx").to(dev))                                      # int32 (B,1)
    assert out.names == ("B", "N", "E")
    assert torch.equal(out.rename(None).cpu(), G(t + "/out"))
    (out.rename(None) * G(t + "/gout").to(dev)).sum().backward()
    g = m.embedding.weight.grad.cpu()
    assert rel_err(g, G(t + "/gweight")) <= TOL32
    assert float(g[0].abs().max()) == 0.0


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_field_aware_embedding_golden(golden, dev, shape):
    from torecsys_amd.inputs import MultiIndicesFieldAwareEmbedding
    G = golden("inputs")
    t = "fa/" + _tag(shape)
    tm = "multi/" + _tag(shape)
    ws = G(t + "/weights")
    N = shape[1]
    m = MultiIndicesFieldAwareEmbedding(embed_size=ws.shape[2], field_sizes=G(tm + "/field_sizes").tolist()).to(dev)
    assert list(m.state_dict().keys()) == [f"embeddings.{i}.weight" for i in range(N)]
    for i, e in enumerate(m.embeddings):
        e.weight.data.copy_(ws[i])
    out = m(G(tm + "/idx").to(dev))
    assert out.names == ("B", "N", "E") and out.shape[1] == N * N
    ref = O.multi_indices_field_aware_embedding(list(ws), G(tm + "/idx"), O.field_offsets(G(tm + "/field_sizes").tolist()))
    assert torch.equal(out.rename(None).cpu(), ref)
    stride = int(G(t + "/out_sub_stride")[0])
    assert torch.equal(out.rename(None).cpu()[:, ::stride], G(t + "/out_sub"))
    out.rename(None).sum().backward()
    assert rel_err(m.embeddings[1].weight.grad.cpu(), G(t + "/gweight1_ones")) <= TOL32


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_fm_layer_golden(golden, dev, shape):
    from torecsys_amd.layers import FactorizationMachineLayer, FMLayer
    assert FMLayer is FactorizationMachineLayer
    G = golden("layers")
    tag = _tag(shape)
    x = G(f"fm/{tag}/x").to(dev).requires_grad_()
    lay = FactorizationMachineLayer(dropout_p=0.0)
    xin = x.refine_names("B", "N", "E")
    y = lay(xin)
    assert y.names == ("B", "O")
    assert rel_err(y.rename(None).cpu(), G(f"fm/{tag}/out")) <= TOL32
    (y.rename(None) * G(f"fm/{tag}/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(f"fm/{tag}/gx")) <= TOL32


def _rand_case(B, N, E, V_per_field, seed, dtype, zipf=False):
    g = torch.Generator().manual_seed(seed)
    fs = [V_per_field + (i % 3) for i in range(N)]
    if zipf:
        cols = []
        for f in fs:
            u = torch.rand(B, 1, generator=g)
            cols.append((f * u ** 6).long().clamp_(0, f - 1))      # heavy head: many repeats of row 0
        idx = torch.cat(cols, 1)
    else:
        idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1)
    w = torch.randn(sum(fs), E, generator=g).to(dtype)
    w1 = torch.randn(sum(fs), 1, generator=g).to(dtype)
    return fs, idx, w, w1, g


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,E,Vf,zipf", [(1024, 10, 16, 100, False), (512, 39, 64, 50, False),
                                           (4096, 39, 64, 7, True), (300, 5, 10, 9, False), (64, 3, 128, 4, False),
                                           (1024, 10, 16, 10000, False)])      # the last: BASELINE configs[0]'s shape
def test_fused_embed_fm_vs_oracle(dev, dtype, B, N, E, Vf, zipf):
    """embed_fm (lookup + FM + first-order sum in one kernel) and its backward (FM term folded into the
    segmented scatter; hot rows through the long-row path) against autograd over the oracle."""
    from torecsys_amd import functional as F_
    fs, idx, w, w1, g = _rand_case(B, N, E, Vf, 77 + B + N + E, dtype, zipf)
    off = O.field_offsets(fs)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    # oracle in fp32 on the (possibly bf16-rounded) weights
    wr = w.float().clone().requires_grad_()
    w1r = w1.float().clone().requires_grad_()
    emb_r = O.multi_indices_embedding(wr, idx, off)
    fm_r = O.fm_layer(emb_r)
    first_r = O.multi_indices_embedding(w1r, idx, off).sum(dim=1)
    ge = torch.randn(B, N, E, generator=g).to(dtype)
    gf = torch.randn(B, E, generator=g).to(dtype)
    g1 = torch.randn(B, 1, generator=g).to(dtype)
    ((emb_r * ge.float()).sum() + (fm_r * gf.float()).sum() + (first_r * g1.float()).sum()).backward()

    wd = w.to(dev).requires_grad_()
    w1d = w1.to(dev).requires_grad_()
    emb, fm, first = F_.embed_fm(wd, idx.to(dev), off.to(dev), w1d)
    assert torch.equal(emb.cpu(), emb_r.detach().to(dtype))                  # gather is bit-exact
    assert rel_err(fm.float().cpu(), fm_r.detach()) <= tol
    assert rel_err(first.float().cpu(), first_r.detach()) <= tol
    ((emb.float() * ge.to(dev).float()).sum() + (fm.float() * gf.to(dev).float()).sum()
     + (first.float() * g1.to(dev).float()).sum()).backward()
    assert rel_err(wd.grad.float().cpu(), wr.grad) <= tol
    assert rel_err(w1d.grad.float().cpu(), w1r.grad) <= tol
    # fm only (no block written), fm unused -> gradient only from the block
    wd2 = w.to(dev).requires_grad_()
    emb2, fm2, _ = F_.embed_fm(wd2, idx.to(dev), off.to(dev), None)
    (emb2.float() * ge.to(dev).float()).sum().backward()
    wr2 = w.float().clone().requires_grad_()
    (O.multi_indices_embedding(wr2, idx, off) * ge.float()).sum().backward()
    assert rel_err(wd2.grad.float().cpu(), wr2.grad) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,E,Vf,zipf", [(1024, 10, 16, 100, False), (512, 39, 64, 50, False),
                                           (4096, 39, 64, 7, True), (8192, 5, 8, 3, True), (64, 3, 128, 4, False)])
def test_embed_fm_fields_vs_oracle(dev, dtype, B, N, E, Vf, zipf):
    """trs_embed_fm_fields / trs_scatter_rows_first: the per-field first-order lookup (B,N,1) rides in the wide table's
    lookup kernel and its gradient in the wide table's bucket walk (hot rows through the long-row kernels); values are
    bit-exact gathers, both tables' gradients against autograd over the oracle with one gradient value per lookup."""
    from torecsys_amd import functional as F_
    fs, idx, w, w1, g = _rand_case(B, N, E, Vf, 177 + B + N + E, dtype, zipf)
    off = O.field_offsets(fs)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    wr = w.float().clone().requires_grad_()
    w1r = w1.float().clone().requires_grad_()
    emb_r = O.multi_indices_embedding(wr, idx, off)
    fm_r = O.fm_layer(emb_r)
    first_r = O.multi_indices_embedding(w1r, idx, off)                         # (B,N,1): not summed
    ge = torch.randn(B, N, E, generator=g).to(dtype)
    gf = torch.randn(B, E, generator=g).to(dtype)
    g1 = torch.randn(B, N, 1, generator=g).to(dtype)
    ((emb_r * ge.float()).sum() + (fm_r * gf.float()).sum() + (first_r * g1.float()).sum()).backward()

    wd = w.to(dev).requires_grad_()
    w1d = w1.to(dev).requires_grad_()
    emb, fm, first = F_.embed_fm_fields(wd, w1d, idx.to(dev), off.to(dev))
    assert first.shape == (B, N, 1)
    assert torch.equal(emb.cpu(), emb_r.detach().to(dtype))
    assert torch.equal(first.cpu(), first_r.detach().to(dtype))
    assert rel_err(fm.float().cpu(), fm_r.detach()) <= tol
    ((emb.float() * ge.to(dev).float()).sum() + (fm.float() * gf.to(dev).float()).sum()
     + (first.float() * g1.to(dev).float()).sum()).backward()
    assert rel_err(wd.grad.float().cpu(), wr.grad) <= tol
    assert rel_err(w1d.grad.float().cpu(), w1r.grad) <= tol
    assert w1d.grad.shape == w1.shape
    # only the first-order output used: the wide table gets no walk of its own, the E = 1 table its own scatter
    wd2, w1d2 = w.to(dev).requires_grad_(), w1.to(dev).requires_grad_()
    _, _, first2 = F_.embed_fm_fields(wd2, w1d2, idx.to(dev), off.to(dev))
    (first2.float() * g1.to(dev).float()).sum().backward()
    assert rel_err(w1d2.grad.float().cpu(), w1r.grad) <= tol
    assert float(wd2.grad.float().abs().max()) == 0.0


def test_inputs_pairs_first_order_table_with_wide_table(dev, monkeypatch):
    """Inputs routes MultiIndicesEmbedding(E, fuse_fm) and MultiIndicesEmbedding(1) over the same columns through ONE
    lookup pass and ONE bucket walk (either schema order); outputs, names and gradients equal the two separate modules.
    A fused optimizer, a padding row or different columns keep the modules separate."""
    from torecsys_amd import functional as F_
    from torecsys_amd import inputs as I_
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    from torecsys_amd.layers import FMLayer
    monkeypatch.setattr(I_, "PAIR_FIRST_ORDER", True)          # opt-in switch (TRS_PAIR_FIRST_ORDER=1)
    fs, idx, w, w1, g = _rand_case(2048, 12, 32, 20, 9, torch.bfloat16, True)
    calls = []
    real = F_.embed_fm_fields

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)

    res = []
    for order, paired in (("ef", True), ("fe", True), ("ef", False)):
        emb = MultiIndicesEmbedding(embed_size=32, field_sizes=fs, fuse_fm=True).to(dev).to(torch.bfloat16)
        feat = MultiIndicesEmbedding(embed_size=1, field_sizes=fs).to(dev).to(torch.bfloat16)
        emb.embedding.weight.data.copy_(w)
        feat.embedding.weight.data.copy_(w1)
        emb.set_schema(["c0"])
        feat.set_schema(["c0"] if paired else ["c1"])
        schema = {"emb_inputs": emb, "feat_inputs": feat} if order == "ef" else {"feat_inputs": feat, "emb_inputs": emb}
        inp = Inputs(schema=schema)
        calls.clear()
        F_.embed_fm_fields = spy
        try:
            d = inp({"c0": idx.to(dev), "c1": idx.to(dev)})
        finally:
            F_.embed_fm_fields = real
        assert len(calls) == (1 if paired else 0)
        assert list(d.keys()) == list(schema.keys())
        assert d["emb_inputs"].names == ("B", "N", "E") and d["feat_inputs"].names == ("B", "N", "E")
        assert hasattr(d["emb_inputs"], "_trs_fused_fm")
        y = FMLayer()(d["emb_inputs"]).rename(None).float().sum(1, keepdim=True) + d["feat_inputs"].rename(None).float().sum(1)
        (y.sum() + (d["emb_inputs"].rename(None).float() ** 2).sum()).backward()
        res.append((d["emb_inputs"].rename(None).detach().cpu(), d["feat_inputs"].rename(None).detach().cpu(),
                    emb.embedding.weight.grad.float().cpu(), feat.embedding.weight.grad.float().cpu()))
    for k in (0, 1):
        assert torch.equal(res[k][0], res[2][0]) and torch.equal(res[k][1], res[2][1])
        assert rel_err(res[k][2], res[2][2]) <= TOLBF and rel_err(res[k][3], res[2][3]) <= TOLBF


@pytest.mark.parametrize("B,N,E,Vf,zipf", [(1024, 10, 32, 100, False), (777, 39, 64, 50, False),
                                           (4096, 39, 64, 7, True), (300, 17, 128, 9, False), (64, 2, 64, 4, False)])
def test_embed_ipn_vs_oracle(dev, B, N, E, Vf, zipf):
    """trs_embed_pair_dot (K7 fused with K1): inner products straight from the table rows, the block written on the way;
    values against the oracle's lookup + inner-product network, table gradient against autograd over the oracle;
    without the block (inference) the same products."""
    from torecsys_amd import functional as F_
    dtype = torch.bfloat16
    fs, idx, w, _, g = _rand_case(B, N, E, Vf, 277 + B + N + E, dtype, zipf)
    off = O.field_offsets(fs)
    wr = w.float().clone().requires_grad_()
    emb_r = O.multi_indices_embedding(wr, idx, off)
    ipn_r = O.inner_product_layer(emb_r)
    P = N * (N - 1) // 2
    ge = torch.randn(B, N, E, generator=g).to(dtype)
    gp = torch.randn(B, P, generator=g).to(dtype)
    ((emb_r * ge.float()).sum() + (ipn_r * gp.float()).sum()).backward()
    assert F_.embed_ipn_supported(w.to(dev), N)
    wd = w.to(dev).requires_grad_()
    emb, ipn = F_.embed_ipn(wd, idx.to(dev), off.to(dev))
    assert torch.equal(emb.cpu(), emb_r.detach().to(dtype))                  # the lookup is bit-exact
    assert rel_err(ipn.float().cpu(), ipn_r.detach()) <= TOLBF
    ((emb.float() * ge.to(dev).float()).sum() + (ipn.float() * gp.to(dev).float()).sum()).backward()
    assert rel_err(wd.grad.float().cpu(), wr.grad) <= 2 * TOLBF              # two bf16 roundings: dx, then the row sums
    with torch.no_grad():
        none, ipn2 = F_.embed_ipn(w.to(dev), idx.to(dev), off.to(dev), want_emb=False)
    assert none is None and torch.equal(ipn2, ipn.detach())
    # an out-of-range id reads as a zero row and is reported
    bad = idx.clone()
    bad[0, 0] = fs[0] + 10 ** 6
    with torch.no_grad():
        _, ipn3 = F_.embed_ipn(w.to(dev), bad.to(dev), off.to(dev), want_emb=False)
    assert float(ipn3[0, :N - 1].float().abs().max()) == 0.0 and F_.index_errors_seen()


def test_fused_ipn_side_channel(dev):
    """MultiIndicesEmbedding(fuse_ipn=True) hands the inner products to InnerProductNetworkLayer without a second pass;
    results and gradients equal the unfused modules (fp32 tables fall back to the plain lookup)."""
    from torecsys_amd.inputs import MultiIndicesEmbedding
    from torecsys_amd.layers import InnerProductNetworkLayer
    fs, idx, w, _, g = _rand_case(512, 12, 32, 20, 15, torch.bfloat16)
    outs = []
    for fuse in (False, True):
        m = MultiIndicesEmbedding(embed_size=32, field_sizes=fs, fuse_ipn=fuse).to(dev).to(torch.bfloat16)
        m.embedding.weight.data.copy_(w)
        emb = m(idx.to(dev))
        assert hasattr(emb, "_trs_fused_ipn") == fuse
        y = InnerProductNetworkLayer(num_fields=12)(emb)
        (y.rename(None).float().sum() + (emb.rename(None).float() ** 2).sum()).backward()
        outs.append((y.rename(None).detach().float().cpu(), m.embedding.weight.grad.float().cpu()))
    assert rel_err(outs[1][0], outs[0][0]) <= TOLBF
    assert rel_err(outs[1][1], outs[0][1]) <= TOLBF
    m32 = MultiIndicesEmbedding(embed_size=32, field_sizes=fs, fuse_ipn=True).to(dev)
    assert not hasattr(m32(idx.to(dev)), "_trs_fused_ipn")


def test_fused_fm_side_channel(dev):
    """MultiIndicesEmbedding(fuse_fm=True) hands the FM term to FMLayer without a second pass; results and
    gradients equal the unfused modules."""
    from torecsys_amd.inputs import MultiIndicesEmbedding
    from torecsys_amd.layers import FMLayer
    fs, idx, w, _, g = _rand_case(256, 12, 32, 20, 5, torch.float32)
    outs = []
    for fuse in (False, True):
        m = MultiIndicesEmbedding(embed_size=32, field_sizes=fs, fuse_fm=fuse).to(dev)
        m.embedding.weight.data.copy_(w)
        emb = m(idx.to(dev))
        assert hasattr(emb, "_trs_fused_fm") == fuse
        y = FMLayer()(emb)
        (y.rename(None).sum() + (emb.rename(None) ** 2).sum()).backward()
        outs.append((y.rename(None).detach().cpu(), m.embedding.weight.grad.cpu()))
    assert rel_err(outs[1][0], outs[0][0]) <= TOL32
    assert rel_err(outs[1][1], outs[0][1]) <= TOL32


def test_gather_properties_full_size(dev):
    """BASELINE shape (B=65536, N=39, E=64, V=1M, bf16): size-independent properties.
    gather == index_select bit-exact; FM is invariant to a permutation of the fields; the dense gradient
    of sum(out) counts the lookups of every row."""
    from torecsys_amd import functional as F_
    B, N, E, V = 65536, 39, 64, 1_000_000
    g = torch.Generator().manual_seed(1234)
    per = V // N
    fs = [per] * (N - 1) + [V - per * (N - 1)]
    off = O.field_offsets(fs).to(dev)
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev)
    w = torch.randn(V, E, generator=g).to(torch.bfloat16).to(dev).requires_grad_()
    emb, fm, _ = F_.embed_fm(w, idx, off, None)
    ref = w.detach()[(idx + off.view(1, -1)).reshape(-1)].reshape(B, N, E)
    assert torch.equal(emb, ref)
    pi = torch.randperm(N, generator=g).to(dev)
    fm_p = F_.fm_layer(ref[:, pi].contiguous())
    assert rel_err(fm_p.float(), fm.float()) <= TOLBF
    s = ref.float().sum(1)
    fm_ref = 0.5 * (s * s - (ref.float() ** 2).sum(1))
    assert rel_err(fm.float(), fm_ref) <= TOLBF
    emb.float().sum().backward()
    counts = torch.bincount((idx + off.view(1, -1)).reshape(-1), minlength=V).float()
    assert torch.equal(w.grad.float()[:, 0], counts.to(torch.bfloat16).float())
    assert torch.equal(w.grad.float()[:, E - 1], counts.to(torch.bfloat16).float())


def test_errors(dev):
    from torecsys_amd import functional as F_
    from torecsys_amd.inputs import MultiIndicesEmbedding
    m = MultiIndicesEmbedding(embed_size=8, field_sizes=[3, 4])
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 2, dtype=torch.long))                 # CPU tensors: no fallback
    m = m.to(dev)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, dtype=torch.long, device=dev))
    with pytest.raises(TypeError):
        F_.gather_rows(m.embedding.weight, torch.zeros(2, 2, device=dev))   # float indices
    with pytest.raises(TypeError):
        F_.gather_rows(m.embedding.weight.half(), torch.zeros(2, 2, dtype=torch.long, device=dev))
    # empty batch
    out = m(torch.zeros(0, 2, dtype=torch.long, device=dev))
    assert out.shape == (0, 2, 8)


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
@pytest.mark.parametrize("dtype,E", [(torch.float32, 16), (torch.float32, 10), (torch.bfloat16, 64)])
def test_fused_sparse_optimizer_equals_dense_step(dev, kind, dtype, E):
    """set_fused_optimizer(): the in-backward row update equals a dense torch.optim step on the dense-gradient path
    (two steps, Zipf-ish indices so hot rows take the long-row kernel), and no .grad is produced."""
    from torecsys_amd.inputs import MultiIndicesEmbedding
    from torecsys_amd.layers import FMLayer
    from torecsys_amd.optim import FusedSparseAdagrad, FusedSparseSGD
    fs, idx0, w, _, g = _rand_case(2048, 7, E, 6, 31 + E, dtype, zipf=True)
    _, idx1, _, _, _ = _rand_case(2048, 7, E, 6, 77 + E, dtype, zipf=True)
    lr = 0.05
    mods = []
    for fused in (False, True):
        m = MultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=(E * w.element_size()) % 16 == 0).to(dev).to(dtype)
        m.embedding.weight.data.copy_(w)
        if fused:
            m.set_fused_optimizer(FusedSparseSGD(lr) if kind == "sgd" else FusedSparseAdagrad(lr, eps=1e-10))
            opt = None
        else:
            opt = (torch.optim.SGD(m.parameters(), lr=lr) if kind == "sgd"
                   else torch.optim.Adagrad(m.parameters(), lr=lr, eps=1e-10))
        for idx in (idx0, idx1):
            emb = m(idx.to(dev))
            y = FMLayer()(emb)
            loss = (y.rename(None).float() ** 2).mean() + emb.rename(None).float().sum() * 1e-3
            if opt is not None:
                opt.zero_grad()
            loss.backward()
            if opt is not None:
                opt.step()
            else:
                assert m.embedding.weight.grad is None
        mods.append(m.embedding.weight.detach().float().cpu())
    tol = 1e-5 if dtype == torch.float32 else 1e-2          # north_star's bounds (measured: 4e-7 / 4.4e-3)
    assert rel_err(mods[1], mods[0]) <= tol
    assert not torch.equal(mods[0], w.float())


@pytest.mark.parametrize("dtype,E", [(torch.float32, 16), (torch.float32, 10), (torch.bfloat16, 64)])
def test_fused_sparse_adam_equals_torch_sparse_adam(dev, dtype, E):
    """FusedSparseAdam (lazy Adam inside the scatter pass) == torch.optim.SparseAdam fed the coalesced sparse
    gradient of the same lookups, three steps; fp32 master copy on the torch side for the bf16 table."""
    from torecsys_amd.inputs import MultiIndicesEmbedding
    from torecsys_amd.layers import FMLayer
    from torecsys_amd.optim import FusedSparseAdam
    fs, idx0, w, _, g = _rand_case(2048, 7, E, 6, 5 + E, dtype, zipf=True)
    batches = [idx0] + [_rand_case(2048, 7, E, 6, 100 + k + E, dtype, zipf=True)[1] for k in range(2)]
    lr, betas, eps = 0.01, (0.9, 0.99), 1e-8
    fuse = (E * w.element_size()) % 16 == 0
    off = torch.tensor([0] + list(torch.tensor(fs).cumsum(0)[:-1]), device=dev)

    def loss_of(m, idx):
        emb = m(idx.to(dev))
        y = FMLayer()(emb)
        return (y.rename(None).float() ** 2).mean() + emb.rename(None).float().sum() * 1e-3

    fused = MultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse).to(dev).to(dtype)
    fused.embedding.weight.data.copy_(w)
    fused.set_fused_optimizer(FusedSparseAdam(lr, betas=betas, eps=eps))
    plain = MultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse).to(dev).to(dtype)
    plain.embedding.weight.data.copy_(w)
    master = torch.nn.Parameter(w.to(dev).float().clone())
    ref_opt = torch.optim.SparseAdam([master], lr=lr, betas=betas, eps=eps)
    for idx in batches:
        loss_of(fused, idx).backward()
        assert fused.embedding.weight.grad is None
        plain.embedding.weight.grad = None
        loss_of(plain, idx).backward()
        rows = (idx.to(dev) + off.view(1, -1)).reshape(-1).unique()
        G = plain.embedding.weight.grad.float()
        master.grad = torch.sparse_coo_tensor(rows.unsqueeze(0), G[rows], size=G.shape)
        ref_opt.step()
        plain.embedding.weight.data.copy_(master.data)      # keep the forward of the reference path in step
    tol = 1e-5 if dtype == torch.float32 else 1e-2          # north_star's bounds (measured: 5e-7 / 5.4e-3)
    assert rel_err(fused.embedding.weight.detach().float().cpu(), master.detach().cpu()) <= tol
    assert not torch.equal(master.detach().cpu(), w.float())


def _check_csr(rb, rows_flat, V):
    """row_start = exclusive prefix sum of the per-row lookup counts; perm lists, row by row, exactly the flat
    lookup positions that hit the row (any order inside a row)."""
    valid = (rows_flat >= 0) & (rows_flat < V)
    counts = torch.bincount(rows_flat[valid], minlength=V)
    expect_start = torch.zeros(V + 1, dtype=torch.int64, device=rows_flat.device)
    expect_start[1:] = counts.cumsum(0)
    assert torch.equal(rb.row_start.long(), expect_start)
    total = int(expect_start[-1])
    perm = rb.perm[:total].long()
    assert torch.equal(perm.sort().values, valid.nonzero().flatten())       # a permutation of the valid lookups
    got_rows = rows_flat[perm]
    assert torch.equal(got_rows, got_rows.sort().values)                    # grouped by destination row
    assert torch.equal(torch.bincount(got_rows, minlength=V), counts)


@pytest.mark.parametrize("case", ["criteo", "tiny_fields", "spill", "unsorted_offsets", "int32", "huge_field", "zipf",
                                  "tiny_tail", "ragged_batch"])
def test_row_buckets_partitioned_and_fallback(dev, case):
    """trs_csr_build: the partitioned LDS-counter build (B >= 2048, per-field ranges) and its device-side fall-back
    to the global-atomic build must produce the same CSR as a bincount/sort restatement."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(11)
    B = 4096
    if case == "tiny_fields":
        sizes = [1, 2, 3, 1, 7, 40000, 5]
    elif case == "huge_field":
        sizes = [100, 70000, 9]          # 5 chunks in one field
    elif case == "tiny_tail":
        sizes = [2053, 1029, 3]          # chunks of 1024 rows: both long fields end in a 5-row chunk (privatised counters)
    elif case == "ragged_batch":
        sizes, B = [2564] * 5 + [17, 1, 333, 20000], 5000 + 37          # batch not a multiple of any kernel's stride
    else:
        sizes = [2564] * 13 + [20000, 17, 1, 333, 15361, 15360, 15359]
    N = len(sizes)
    V = sum(sizes)
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)[:-1]), dtype=torch.int64)
    if case == "zipf":
        idx = torch.stack([(torch.rand(B, generator=g) ** 6 * s).long().clamp_(max=s - 1) for s in sizes], 1)
    else:
        idx = torch.stack([torch.randint(0, s, (B,), generator=g) for s in sizes], 1)
    if case == "spill":                   # legal for nn.Embedding: idx + offset < V but outside the field's own range
        idx[5, 0] = sizes[0] + 3
        idx[77, 3] = 40
    if case == "unsorted_offsets":
        off = off.flip(0).contiguous()
        idx = idx.flip(1).contiguous()
    if case == "int32":
        idx = idx.int()
    idx, off = idx.to(dev), off.to(dev)
    F_.clear_caches()
    rb = F_.row_buckets(idx, off, V)
    torch.cuda.synchronize()
    rows_flat = (idx.long() + off.view(1, -1)).reshape(-1)
    _check_csr(rb, rows_flat, V)
    # and the gradient built from it
    w = torch.randn(V, 16, device=dev, requires_grad=True)
    out = F_.gather_rows(w, idx, off)
    gout = torch.randn_like(out)
    out.backward(gout)
    ref = torch.zeros(V, 16, device=dev).index_add_(0, rows_flat, gout.reshape(-1, 16))
    assert rel_err(w.grad.cpu(), ref.cpu()) <= 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("fuse", [False, True])
def test_very_hot_rows_are_chunked(dev, dtype, tol, fuse):
    """A row that collects thousands of lookups is cut into 1024-lookup chunks reduced by different workgroups and
    finished by a second pass (Zipf head rows): same gradient as index_add, with and without the folded FM term."""
    from torecsys_amd.inputs import MultiIndicesEmbedding
    from torecsys_amd.layers import FMLayer
    g = torch.Generator().manual_seed(5)
    B, E = 5000, 64
    fs = [50, 7, 300]
    idx = torch.stack([torch.randint(0, s, (B,), generator=g) for s in fs], 1)
    idx[:, 0] = 3                          # 5000 lookups of one row -> 5 chunks
    idx[: 2500, 1] = 1                     # 2500+ lookups -> 3 chunks
    idx[: 1100, 2] = 42                    # just above one chunk
    w = torch.randn(sum(fs), E, generator=g)
    m = MultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse).to(dev).to(dtype)
    m.embedding.weight.data.copy_(w)
    emb = m(idx.to(dev))
    y = FMLayer()(emb)
    gy = torch.randn(B, E, generator=g)
    ge = torch.randn(B, len(fs), E, generator=g)
    ((y.rename(None).float() * gy.to(dev)).sum() + (emb.rename(None).float() * ge.to(dev)).sum()).backward()
    wr = m.embedding.weight.detach().float().cpu().requires_grad_()
    off = O.field_offsets(fs)
    er = O.multi_indices_embedding(wr, idx, off)
    ((O.fm_layer(er) * gy).sum() + (er * ge).sum()).backward()
    assert rel_err(m.embedding.weight.grad.float().cpu(), wr.grad) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("E,bcast", [(1, True), (1, False), (10, False)])
def test_element_path_splits_very_hot_rows_over_waves(dev, dtype, tol, E, bcast):
    """criteo-skewed fields (4, 5, 6 ... rows): a row of the E = 1 first-order table (or of any table whose rows are not
    whole 16-byte vectors) collects thousands of lookups; above 2048 it is cut into chunks reduced by different waves
    whose partial sums meet in fp32 accumulators (round 4: one wave per row took 115 us of the skewed DeepFM step).  Also
    the row-bucket build on fields of <= 32 rows (wave-aggregated LDS atomics).  Same gradient as index_add."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(31 + E)
    B, fs = 20000, [4, 5, 2, 33, 700, 9]
    N, V = len(fs), sum(fs)
    off = O.field_offsets(fs)
    idx = torch.stack([torch.randint(0, f, (B,), generator=g) for f in fs], 1)
    idx[:, 2] = 1                                                     # 20 000 lookups of one row: 10 chunks
    idx[:2100, 4] = 77                                                # just above one chunk
    idx[:2048, 5] = 3                                                 # exactly the limit: stays on the one-wave path
    w = torch.randn(V, E, generator=g).to(dtype)
    wd = w.to(dev).requires_grad_()
    out = F_.gather_rows(wd, idx.to(dev), off.to(dev))
    rows = (idx + off.view(1, -1)).reshape(-1)
    assert torch.equal(out.detach().cpu(), w[rows].reshape(B, N, E))
    if bcast:                         # the models' first-order term: one gradient row per sample, shared by its fields
        gs = torch.randn(B, 1, E, generator=g).to(dtype)
        out.backward(gs.to(dev).expand(B, N, E))
        gfull = gs.float().expand(B, N, E)
    else:
        gfull = torch.randn(B, N, E, generator=g).to(dtype)
        out.backward(gfull.to(dev))
        gfull = gfull.float()
    ref = torch.zeros(V, E, dtype=torch.float64).index_add_(0, rows, gfull.reshape(-1, E).double())
    assert rel_err(wd.grad.double().cpu(), ref) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("E", [1, 10, 64])
def test_gather_backward_of_a_field_constant_gradient(dev, dtype, tol, E):
    """out.sum over the fields (the models' first-order term, models/ctr/deep_fm.py:55-110) feeds back the same row for
    every field of a sample -- an expanded (B,1,E) gradient.  The gather backward reads it as one row per sample
    (broadcast mode of trs_scatter_rows) instead of materialising (B,N,E): same gradient as index_add on the expanded
    block, for the E = 1 element path, the any-E element path, the vector path and the padding row."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(99 + E)
    B, fs = 3000, [40, 7, 300, 5]
    N, V = len(fs), sum(fs)
    off = O.field_offsets(fs)
    idx = torch.stack([torch.randint(0, f, (B,), generator=g) for f in fs], 1)
    idx[:2000, 1] = 2                                                # a hot row (long-row queue)
    w = torch.randn(V, E, generator=g).to(dtype)
    gs = torch.randn(B, 1, E, generator=g).to(dtype)
    pad = int(off[2]) + 5
    wd = w.to(dev).requires_grad_()
    out = F_.gather_rows(wd, idx.to(dev), off.to(dev), padding_idx=pad)
    gexp = gs.to(dev).expand(B, N, E)
    assert gexp.stride(1) == 0
    out.backward(gexp)
    rows = (idx + off.view(1, -1)).reshape(-1)
    ref = torch.zeros(V, E).index_add_(0, rows, gs.float().expand(B, N, E).reshape(-1, E))
    ref[pad] = 0
    assert rel_err(wd.grad.float().cpu(), ref) <= tol
    wd2 = w.to(dev).requires_grad_()                                  # the materialised block takes the ordinary path
    F_.gather_rows(wd2, idx.to(dev), off.to(dev), padding_idx=pad).backward(gexp.contiguous())
    assert rel_err(wd.grad.float().cpu(), wd2.grad.float().cpu()) <= (1e-6 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,E,Vf,hot", [(1024, 10, 16, 100, False), (4096, 39, 64, 7, True), (300, 5, 10, 9, False),
                                           (512, 6, 4, 50, False), (512, 6, 8, 50, True)])     # rows of ONE 16-byte vector
def test_fm_gradient_constant_along_E(dev, dtype, tol, B, N, E, Vf, hot):
    """The reference's FM / DeepFM models sum the FM output over E (models/ctr/deep_fm.py:55-110), so the gradient that
    reaches the fused lookup+FM backward is an expanded (B,1) column.  trs_scatter_rows reads it as one value per sample
    (g_fm_cols = 1): same table gradient as the oracle, and as the full-row path on the materialised gradient -- through
    the bucket walk, the hot-row queue (``hot``), the any-E element path (E = 10) and the fused optimizer."""
    from torecsys_amd import functional as F_
    from torecsys_amd.optim import FusedSparseSGD
    fs, idx, w, _, g = _rand_case(B, N, E, Vf, 321 + B + E, dtype, hot)
    off = O.field_offsets(fs)
    ge = torch.randn(B, N, E, generator=g).to(dtype)
    g1 = torch.randn(B, 1, generator=g).to(dtype)
    wr = w.float().clone().requires_grad_()
    emb_r = O.multi_indices_embedding(wr, idx, off)
    ((emb_r * ge.float()).sum() + (O.fm_layer(emb_r).sum(dim=1, keepdim=True) * g1.float()).sum()).backward()

    def run(expand, opt=None):
        wd = w.to(dev).clone().requires_grad_()
        emb, fm, _ = F_.embed_fm(wd, idx.to(dev), off.to(dev), None, True, opt)
        gfm = g1.to(dev).expand(B, E)
        gfm = gfm if expand else gfm.contiguous()
        torch.autograd.backward([emb, fm], [ge.to(dev), gfm])
        return wd
    assert F_._fm_grad_operand(g1.to(dev).expand(B, E)).shape == (B, 1)
    got = run(True).grad.float().cpu()
    assert rel_err(got, wr.grad) <= tol
    full = run(False).grad.float().cpu()
    assert rel_err(got, full) <= (1e-6 if dtype == torch.float32 else 8e-3)      # same sums, g*S rounded once either way
    wd = run(True, FusedSparseSGD(0.5))                                           # applied in place by the fused optimizer
    assert wd.grad is None
    assert rel_err(wd.detach().float().cpu(), w.float() - 0.5 * wr.grad) <= tol


def test_out_of_range_lookup_is_sanitised_and_reported_lazily(dev):
    """without TRS_CHECK_INDICES the kernels still range-check: an out-of-range lookup reads as a zero row, gets no
    gradient, and functional.index_errors_seen() reports it afterwards (no host read-back inside the lookup)"""
    from torecsys_amd import functional as F_
    from torecsys_amd.inputs import MultiIndicesEmbedding
    if F_.CHECK_INDICES:
        pytest.skip("TRS_CHECK_INDICES=1 raises at the call instead")
    torch.manual_seed(0)
    emb = MultiIndicesEmbedding(embed_size=16, field_sizes=[5, 7, 11]).to(dev)
    F_.index_errors_seen()                                  # clear
    idx = torch.tensor([[1, 2, 3], [4, 6, 10]], device=dev)
    out = emb(idx).rename(None)
    assert not F_.index_errors_seen()
    bad = torch.tensor([[1, 2, 3], [4, 6, 11]], device=dev)  # field 2 has 11 rows: 11 + offset 12 = 23 = V -> outside
    out_bad = emb(bad).rename(None)
    assert torch.equal(out_bad[0], out[0]) and torch.equal(out_bad[1, :2], out[1, :2])
    assert torch.count_nonzero(out_bad[1, 2]) == 0
    out_bad.sum().backward()
    assert torch.isfinite(emb.embedding.weight.grad).all()
    assert F_.index_errors_seen() and not F_.index_errors_seen()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("pad", [0, 7, -1])
def test_fused_lookup_fm_respects_padding_idx(dev, dtype, tol, pad):
    """multi_indices_emb.py:48 forwards ``padding_idx`` to nn.Embedding: that row is read like any other and gets NO
    gradient.  The fused lookup + FM path (fuse_fm=True) must behave like F.embedding(padding_idx=) + FM: same block
    (bit-exact), same FM term, same dense gradient with an exactly-zero padding row -- also through the fused optimizer."""
    from torecsys_amd.inputs import MultiIndicesEmbedding
    from torecsys_amd.layers import FactorizationMachineLayer
    from torecsys_amd.optim import FusedSparseSGD
    B, E, fs = 700, 16, [5, 9, 3, 11]
    V = sum(fs)
    g = torch.Generator().manual_seed(11 + (pad % V))
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1)
    off = O.field_offsets(fs)
    prow = pad % V
    assert bool(((idx + off) == prow).any())                            # the padding row IS looked up
    torch.manual_seed(5)
    mod = MultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=True, padding_idx=pad).to(dev).to(dtype)
    assert mod.padding_idx == prow
    w0 = torch.randn(V, E, generator=g).to(dtype)
    mod.embedding.weight.data.copy_(w0)
    fm = FactorizationMachineLayer()
    ge = torch.randn(B, len(fs), E, generator=g).to(dtype)
    g1 = torch.randn(B, 1, generator=g).to(dtype)
    # reference semantics on the CPU: F.embedding with padding_idx, dense gradient
    wr = w0.float().clone().requires_grad_()
    emb_r = torch.nn.functional.embedding(idx + off, wr, padding_idx=prow)
    ((emb_r * ge.float()).sum() + (O.fm_layer(emb_r).sum(1, keepdim=True) * g1.float()).sum()).backward()
    assert float(wr.grad[prow].abs().max()) == 0.0
    out = mod(idx.to(dev))
    assert hasattr(out, "_trs_fused_fm")
    y = fm(out)
    assert torch.equal(out.rename(None).float().cpu(), emb_r.detach())
    ((out.rename(None) * ge.to(dev)).sum() + (y.rename(None).sum(1, keepdim=True) * g1.to(dev)).sum()).backward()
    got = mod.embedding.weight.grad.float().cpu()
    assert float(got[prow].abs().max()) == 0.0
    assert rel_err(got, wr.grad) <= tol
    # fused optimizer: the padding row keeps its value
    mod.embedding.weight.grad = None
    mod.set_fused_optimizer(FusedSparseSGD(0.25))
    out = mod(idx.to(dev))
    ((out.rename(None) * ge.to(dev)).sum() + (fm(out).rename(None).sum(1, keepdim=True) * g1.to(dev)).sum()).backward()
    wn = mod.embedding.weight.detach().float().cpu()
    assert torch.equal(wn[prow], w0[prow].float())
    assert rel_err(wn, w0.float() - 0.25 * wr.grad) <= tol


@pytest.mark.parametrize("kind", ["adagrad", "adam"])
def test_fused_optimizer_state_of_first_order_table_persists(dev, kind):
    """F_.embed_fm(first_weight=..., opt=...) steps the E = 1 first-order table through a reshaped VIEW of the parameter;
    its optimizer state must be filed under the parameter (one entry, accumulating over the steps), not under the
    per-call view: three steps against torch.optim on the dense gradients."""
    from torecsys_amd import functional as F_
    from torecsys_amd.optim import FusedSparseAdagrad, FusedSparseAdam
    B, N, E, fs = 400, 4, 16, [6, 4, 9, 5]
    V = sum(fs)
    g = torch.Generator().manual_seed(3)
    off = O.field_offsets(fs)
    w0, f0 = torch.randn(V, E, generator=g), torch.randn(V, 1, generator=g)
    wd, fd = w0.to(dev).requires_grad_(), f0.to(dev).requires_grad_()
    wr, fr = w0.clone().requires_grad_(), f0.clone().requires_grad_()
    if kind == "adagrad":
        opt = FusedSparseAdagrad(lr=0.1, eps=1e-10)
        ref = torch.optim.Adagrad([wr, fr], lr=0.1, eps=1e-10)
    else:
        opt = FusedSparseAdam(lr=0.05)
        ref = torch.optim.SparseAdam([wr, fr], lr=0.05)
    for step in range(3):
        idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1)
        emb, fm, first = F_.embed_fm(wd, idx.to(dev), off.to(dev), fd, True, opt)
        (fm.sum() + first.sum() * 0.5 + emb.sum() * 0.1).backward()
        assert wd.grad is None and fd.grad is None
        ref.zero_grad()
        rows = idx + off
        if kind == "adagrad":
            emb_r = wr[rows]
            (O.fm_layer(emb_r).sum() + fr[rows].sum() * 0.5 + emb_r.sum() * 0.1).backward()
        else:       # SparseAdam wants sparse gradients: build them from the dense ones
            emb_r = wr[rows]
            (O.fm_layer(emb_r).sum() + fr[rows].sum() * 0.5 + emb_r.sum() * 0.1).backward()
            touched = torch.unique(rows)
            for p_ in (wr, fr):
                p_.grad = torch.sparse_coo_tensor(touched.unsqueeze(0), p_.grad[touched], p_.shape).coalesce()
        ref.step()
        assert len(opt._state) == 2, "one state entry per table, not one per step"
        assert rel_err(wd.detach().cpu(), wr.detach()) <= 1e-5, step
        assert rel_err(fd.detach().cpu(), fr.detach()) <= 1e-5, step


def test_stacked_single_index_embeddings_take_one_launch(monkeypatch):
    """A StackedInput of SingleIndexEmbeddings (reference inputs/base/stacked_inp.py:94-134: N nn.Embedding lookups + a cat;
    how the reference's own tests and most user code build their inputs) routed by ``Inputs`` onto ONE lookup launch over
    the N separate tables (trs_gather_rows_tables) and one bucket walk: output bit-exact against the oracle's
    single_index_embedding per table, every table's gradient against the oracle's, and equal to the child-by-child path."""
    from oracle import cpu_ref as O
    from torecsys_amd import functional as F_
    from torecsys_amd import inputs as I
    dev = torch.device("cuda:0")

    class StackedInput(I.BaseInput):      # stand-in for the reference's class (Inputs dispatches on the class NAME)
        def __init__(self, inputs):
            super().__init__()
            self.inputs = inputs
            self.length = len(inputs[0])
            for i, inp in enumerate(inputs):
                self.add_module(f"Input_{i}", inp)
            self.set_schema([c for inp in inputs for c in inp.schema.inputs])

        def forward(self, d):
            outs = []
            for inp in self.inputs:
                v = d[inp.schema.inputs[0]]
                outs.append(inp(v.unsqueeze(-1) if v.dim() == 1 else v).rename(None))
            out = torch.cat(outs, dim=1)
            out.names = ("B", "N", "E")
            return out

    g = torch.Generator().manual_seed(11)
    for dt, E, B, sizes, idt in ((torch.float32, 16, 300, [7, 1, 50, 3, 1000], torch.int64),
                                 (torch.bfloat16, 64, 4096, [40 + 13 * i for i in range(39)], torch.int32),
                                 (torch.float32, 10, 257, [5, 9, 2], torch.int64)):      # rows that are not 16-byte multiples
        children = []
        for i, V in enumerate(sizes):
            c = I.SingleIndexEmbedding(E, V)
            c.set_schema([f"c{i}"])
            children.append(c)
        router = I.Inputs({"emb_inputs": StackedInput(children)}).to(dev).to(dt)
        cols = {f"c{i}": torch.randint(0, V, (B,), generator=g).to(idt).to(dev) for i, V in enumerate(sizes)}
        gout = torch.randn(B, len(sizes), E, generator=g).to(dt).to(dev)
        calls = {"tables": 0, "rows": 0}
        real_t, real_r = F_.gather_rows_tables, F_.gather_rows
        monkeypatch.setattr(F_, "gather_rows_tables", lambda *a, **k: (calls.__setitem__("tables", calls["tables"] + 1), real_t(*a, **k))[1])
        monkeypatch.setattr(F_, "gather_rows", lambda *a, **k: (calls.__setitem__("rows", calls["rows"] + 1), real_r(*a, **k))[1])
        out = router(cols)["emb_inputs"]
        assert calls == {"tables": 1, "rows": 0}, calls
        assert out.names == ("B", "N", "E") and tuple(out.shape) == (B, len(sizes), E)
        out.rename(None).backward(gout)
        torch.cuda.synchronize()
        got = [c.embedding.weight.grad.clone() for c in children]
        # oracle: one single_index_embedding per table on the CPU
        ws = [c.embedding.weight.detach().cpu().float().requires_grad_() for c in children]      # (exact images of the bf16 rows)
        ref = torch.cat([O.single_index_embedding(w, cols[f"c{i}"].cpu().long().unsqueeze(-1), None) for i, w in enumerate(ws)], 1)
        assert torch.equal(out.rename(None).detach().float().cpu(), ref.detach()), "lookup not bit-exact"
        ref.backward(gout.float().cpu())
        tol = 1e-5 if dt == torch.float32 else 1e-2
        for i, (a, w) in enumerate(zip(got, ws)):
            assert rel_err(a.float().cpu(), w.grad.float()) <= tol, i
        # ... and the child-by-child path gives the same
        for c in children:
            c.embedding.weight.grad = None
        monkeypatch.setattr(I, "STACKED_ONE_LAUNCH", False)
        out2 = router(cols)["emb_inputs"]
        assert calls["tables"] == 1 and calls["rows"] == len(sizes)
        assert torch.equal(out2.rename(None), out.rename(None))
        monkeypatch.setattr(I, "STACKED_ONE_LAUNCH", True)
        monkeypatch.setattr(F_, "gather_rows_tables", real_t)
        monkeypatch.setattr(F_, "gather_rows", real_r)
    # an index outside its table raises the device-side flag
    bad = dict(cols)
    bad["c1"] = torch.full_like(cols["c1"], sizes[1])
    router(bad)
    assert F_.index_errors_seen(dev)


def test_fused_lookup_fm_on_a_table_beyond_the_caches_streams_its_rows():
    """A table larger than 512 MiB takes the kernel instantiation that fetches rows with streaming loads: same results --
    block bit-exact against index_select, FM term against the fp32 formula on the same rows."""
    from torecsys_amd import functional as F_
    dev = torch.device("cuda:0")
    V, E, N, B = 5_000_000, 64, 13, 4096
    assert V * E * 2 > (512 << 20)
    g = torch.Generator(device=dev).manual_seed(3)
    w = torch.randn(V, E, generator=g, device=dev).bfloat16()
    per = V // N
    sizes = [per] * (N - 1) + [V - per * (N - 1)]
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.tensor(sizes), 0)[:-1]]).to(dev)
    idx = torch.stack([torch.randint(0, s_, (B,), generator=g, device=dev) for s_ in sizes], 1)
    emb, fm, _ = F_.embed_fm(w, idx, off)
    rows = w.index_select(0, (idx + off).reshape(-1)).reshape(B, N, E)
    assert torch.equal(emb, rows)
    assert torch.equal(F_.gather_rows(w, idx, off), rows)      # the plain lookup streams such a table's rows as well
    x = rows.float()
    ref = 0.5 * (x.sum(1) ** 2 - (x * x).sum(1))
    assert rel_err(fm.float().cpu(), ref.cpu()) <= 1e-2


@pytest.mark.parametrize("prefetch", [False, True])
def test_shared_row_buckets_across_streams_inline_builds_and_three_index_sets(dev, monkeypatch, prefetch):
    """A row-bucket entry is shared by every table looked up with the same indices (DeepFM's wide table and its E = 1
    first-order table), and those lookups' backwards can run on two streams (Inputs puts lookups beyond the first on
    the "lookup" stream).  With prefetching off -- or a prefetched entry evicted from the two-entry cache by a third
    index set -- the build happens INLINE in whichever backward comes first: the entry must carry its event so the walk
    on the other stream waits for it.  Three index sets alive at once, gradients against autograd over the oracle."""
    from torecsys_amd import functional as F_
    from torecsys_amd import inputs as I_
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    monkeypatch.setattr(F_, "PREFETCH_BUCKETS", prefetch)
    monkeypatch.setattr(I_, "LOOKUP_STREAMS", 1)
    F_.clear_caches()
    B, N, E = 8192, 39, 64
    fs, _, w, w1, g = _rand_case(B, N, E, 300, 5, torch.float32)
    off = O.field_offsets(fs)
    emb = MultiIndicesEmbedding(embed_size=E, field_sizes=fs).to(dev)
    feat = MultiIndicesEmbedding(embed_size=1, field_sizes=fs).to(dev)
    emb.embedding.weight.data.copy_(w)
    feat.embedding.weight.data.copy_(w1)
    emb.set_schema(["c0"])
    feat.set_schema(["c0"])
    inp = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat})
    sets = [torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1) for _ in range(3)]
    dsets = [s.to(dev) for s in sets]
    ge = torch.randn(B, N, E, generator=g)
    gf = torch.randn(B, N, 1, generator=g)
    for rnd in range(3):
        # three forwards first (three index sets: the two-entry cache evicts the first one), then their backwards
        outs = [inp({"c0": d}) for d in dsets]
        for k, d in enumerate(outs):
            for p in (emb.embedding.weight, feat.embedding.weight):
                p.grad = None
            loss = ((d["emb_inputs"].rename(None) * ge.to(dev)).sum() + (d["feat_inputs"].rename(None) * gf.to(dev)).sum())
            loss.backward()
            torch.cuda.synchronize()
            wr, w1r = w.clone().requires_grad_(), w1.clone().requires_grad_()
            ((O.multi_indices_embedding(wr, sets[k], off) * ge).sum()
             + (O.multi_indices_embedding(w1r, sets[k], off) * gf).sum()).backward()
            assert rel_err(emb.embedding.weight.grad.cpu(), wr.grad) <= TOL32, (rnd, k)
            assert rel_err(feat.embedding.weight.grad.cpu(), w1r.grad) <= TOL32, (rnd, k)
    F_.clear_caches()
