"""GPU parity of the SURVEY.md 8f N3 layers (OuterProductNetwork, BilinearInteraction, AFM) through the C ABI:
fp32 against golden vectors captured from the reference (tests/golden/pairs.npz), bf16 and other shapes against the
CPU oracle on the same inputs."""
import pytest
import torch

from conftest import PAIR_SHAPES, pair_heavy_ok, rel_err_both as rel_err, rel_err_rows      # rel_err here = max norm AND per-row norm (conftest.rel_err_both)
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def _tag(s):
    return "%d_%d_%d" % s


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("kt", ["mat", "vec", "num"])
@pytest.mark.parametrize("shape", PAIR_SHAPES)
def test_outer_product_golden(golden, dev, shape, kt):
    from torecsys_amd.layers import OuterProductNetworkLayer
    G = golden("pairs")
    B, N, E = shape
    t = _tag(shape)
    if kt == "mat" and not pair_heavy_ok(N, E):
        pytest.skip("no golden vector for this size")
    lay = OuterProductNetworkLayer(embed_size=E, num_fields=N, kernel_type=kt).to(dev)
    assert tuple(lay.kernel.shape) == tuple(G(f"opn_{kt}/{t}/kernel").shape)
    lay.kernel.data.copy_(G(f"opn_{kt}/{t}/kernel"))
    x = G(f"x/{t}").to(dev).requires_grad_()
    y = lay(x.refine_names('B', 'N', 'E'))
    assert list(y.names) == G(f"opn_{kt}/{t}/names")
    assert rel_err(y.rename(None).cpu(), G(f"opn_{kt}/{t}/out")) <= 1e-5
    (y.rename(None) * G(f"opn_{kt}/{t}/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(f"opn_{kt}/{t}/gx")) <= 1e-5
    assert rel_err(lay.kernel.grad.cpu(), G(f"opn_{kt}/{t}/gkernel")) <= 1e-5
    assert list(lay.state_dict().keys()) == ["kernel"]


@pytest.mark.parametrize("bt", ["all", "each"])
@pytest.mark.parametrize("shape", PAIR_SHAPES)
def test_bilinear_golden(golden, dev, shape, bt):
    from torecsys_amd.layers import BilinearInteractionLayer
    G = golden("pairs")
    B, N, E = shape
    t = _tag(shape)
    if bt == "each" and not pair_heavy_ok(N, E):
        pytest.skip("no golden vector for this size")
    lay = BilinearInteractionLayer(embed_size=E, num_fields=N, bilinear_type=bt).to(dev)
    lay.bilinear.weight.data.copy_(G(f"bil_{bt}/{t}/W"))
    lay.bilinear.bias.data.copy_(G(f"bil_{bt}/{t}/b"))
    x = G(f"x/{t}").to(dev).requires_grad_()
    y = lay(x.refine_names('B', 'N', 'E'))
    assert list(y.names) == G(f"bil_{bt}/{t}/names")
    assert rel_err(y.rename(None).cpu(), G(f"bil_{bt}/{t}/out")) <= 1e-5
    (y.rename(None) * G(f"bil_{bt}/{t}/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(f"bil_{bt}/{t}/gx")) <= 1e-5
    assert rel_err(lay.bilinear.weight.grad.cpu(), G(f"bil_{bt}/{t}/gW")) <= 1e-5
    assert rel_err(lay.bilinear.bias.grad.cpu(), G(f"bil_{bt}/{t}/gb")) <= 1e-5
    assert sorted(lay.state_dict().keys()) == ["bilinear.bias", "bilinear.weight"]


def test_pair_layer_constructor_errors(dev):
    from torecsys_amd.layers import BilinearInteractionLayer, OuterProductNetworkLayer
    with pytest.raises(ValueError):
        OuterProductNetworkLayer(8, 4, "cube")
    with pytest.raises(NotImplementedError):
        BilinearInteractionLayer(8, 4, "interaction")
    with pytest.raises(ValueError):
        BilinearInteractionLayer(8, 4, "none")
    with pytest.raises(RuntimeError):
        BilinearInteractionLayer(8, 4, "all", bias=False)
    lay = OuterProductNetworkLayer(8, 4, "vec").to(dev)
    with pytest.raises(ValueError):
        lay(torch.zeros(2, 5, 8, device=dev))
    with pytest.raises(RuntimeError):
        OuterProductNetworkLayer(8, 4, "vec")(torch.zeros(2, 4, 8))       # CPU tensors: no CPU path


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,E", [(37, 39, 64), (5, 2, 16), (130, 7, 24), (64, 5, 128), (1, 3, 8), (96, 39, 64),
                                   (70, 12, 16), (257, 2, 64)])
def test_pair_layers_vs_oracle(dev, dtype, tol, B, N, E):
    from torecsys_amd.layers import BilinearInteractionLayer, OuterProductNetworkLayer
    if dtype == torch.bfloat16 and E % 8 != 0:
        pytest.skip("bf16 rows must be 16-byte multiples for the pair-product kernel")
    g = torch.Generator().manual_seed(B * 7 + N * 3 + E)
    P = N * (N - 1) // 2
    x0 = (torch.randn(B, N, E, generator=g) * 0.5).to(dtype)
    cases = [("opn", kt) for kt in ("mat", "vec", "num")] + [("bil", bt) for bt in ("all", "each")]
    for fam, kind in cases:
        torch.manual_seed(5)
        if fam == "opn":
            lay = OuterProductNetworkLayer(E, N, kind).to(dev).to(dtype)
            params = [lay.kernel]
        else:
            lay = BilinearInteractionLayer(E, N, kind).to(dev).to(dtype)
            params = [lay.bilinear.weight, lay.bilinear.bias]
        x = x0.to(dev).requires_grad_()
        y = lay(x).rename(None)
        xr = x0.float().clone().requires_grad_()
        pr = [p.detach().float().cpu().requires_grad_() for p in params]
        yr = O.outer_product_layer(xr, pr[0], kind) if fam == "opn" else O.bilinear_layer(xr, pr[0], pr[1], kind)
        assert y.shape == yr.shape
        assert rel_err(y.float().cpu(), yr) <= tol, (fam, kind)
        assert rel_err_rows(y.float().cpu(), yr.detach(), floor_frac=5e-2) <= 2 * tol, (fam, kind, "per sample")
        go = torch.randn(yr.shape, generator=g)
        (y.float() * go.to(dev)).sum().backward()
        (yr * go).sum().backward()
        assert rel_err(x.grad.float().cpu(), xr.grad) <= tol, (fam, kind, "gx")
        assert rel_err_rows(x.grad.float().cpu(), xr.grad, floor_frac=5e-2) <= 2 * tol, (fam, kind, "gx per sample")
        for p, r in zip(params, pr):
            assert rel_err(p.grad.float().cpu(), r.grad) <= tol, (fam, kind, "gparam")


def _afm_layer(dev, E, N, A, W1, b1, W2, b2, dtype=torch.float32):
    from torecsys_amd.layers import AttentionalFactorizationMachineLayer
    lay = AttentionalFactorizationMachineLayer(embed_size=E, num_fields=N, attn_size=A, dropout_p=0.0).to(dev).to(dtype)
    lay.attention.Linear.weight.data.copy_(W1)
    lay.attention.Linear.bias.data.copy_(b1)
    lay.attention.OutProj.weight.data.copy_(W2)
    lay.attention.OutProj.bias.data.copy_(b2)
    return lay


@pytest.mark.parametrize("shape", PAIR_SHAPES)
def test_afm_golden(golden, dev, shape):
    G = golden("pairs")
    B, N, E = shape
    t = _tag(shape)
    A = G(f"afm/{t}/W1").shape[0]
    lay = _afm_layer(dev, E, N, A, *[G(f"afm/{t}/{n}") for n in ("W1", "b1", "W2", "b2")])
    x = G(f"afm/{t}/x").to(dev).requires_grad_()
    y, attn = lay(x.refine_names('B', 'N', 'E'))
    assert [str(n) for n in y.names] == G(f"afm/{t}/names") and not attn.has_names()
    assert tuple(attn.shape) == (B, N * (N - 1) // 2, 1)
    assert rel_err(y.rename(None).cpu(), G(f"afm/{t}/out")) <= 1e-5
    assert rel_err(attn.cpu(), G(f"afm/{t}/attn")) <= 1e-5
    ((y.rename(None) * G(f"afm/{t}/gout").to(dev)).sum() + (attn * G(f"afm/{t}/gattn").to(dev)).sum()).backward()
    assert rel_err(x.grad.cpu(), G(f"afm/{t}/gx")) <= 1e-5
    a = lay.attention
    for p, n in ((a.Linear.weight, "gW1"), (a.Linear.bias, "gb1"), (a.OutProj.weight, "gW2")):
        assert rel_err(p.grad.cpu(), G(f"afm/{t}/{n}")) <= 1e-5, n
    # d/d(b2) of a softmax over the logits is identically zero (a shift of every logit): the reference's value is
    # summation noise, so it is compared on an absolute scale
    assert float((a.OutProj.bias.grad.cpu() - G(f"afm/{t}/gb2")).abs().max()) <= 1e-5
    assert sorted(lay.state_dict().keys()) == ["attention.Linear.bias", "attention.Linear.weight",
                                               "attention.OutProj.bias", "attention.OutProj.weight"]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,E,A", [(37, 39, 64, 64), (5, 2, 16, 8), (130, 7, 24, 40), (9, 5, 128, 100), (1, 3, 8, 1),
                                     (64, 10, 32, 16), (33, 12, 128, 32), (20, 39, 64, 128), (3, 2, 64, 48),
                                     (40, 9, 32, 64), (17, 6, 64, 96), (9, 4, 32, 128), (300, 2, 64, 32), (70, 34, 64, 32)])
def test_afm_vs_oracle(dev, dtype, tol, B, N, E, A):
    g = torch.Generator().manual_seed(B + N + E + A)
    x0 = (torch.randn(B, N, E, generator=g) * 0.7).to(dtype)
    ps = [(torch.randn(A, E, generator=g) / E ** 0.5).to(dtype), (torch.randn(A, generator=g) * 0.1).to(dtype),
          (torch.randn(1, A, generator=g) / A ** 0.5).to(dtype), (torch.randn(1, generator=g) * 0.1).to(dtype)]
    lay = _afm_layer(dev, E, N, A, *ps, dtype=dtype)
    x = x0.to(dev).requires_grad_()
    y, attn = lay(x)
    xr = x0.float().clone().requires_grad_()
    pr = [p.float().clone().requires_grad_() for p in ps]
    yr, ar = O.afm_layer(xr, *pr)
    assert rel_err(y.rename(None).float().cpu(), yr) <= tol
    assert rel_err(attn.float().cpu(), ar) <= tol
    go, ga = torch.randn(B, E, generator=g), torch.randn(ar.shape, generator=g)
    ((y.rename(None).float() * go.to(dev)).sum() + (attn.float() * ga.to(dev)).sum()).backward()
    ((yr * go).sum() + (ar * ga).sum()).backward()
    assert rel_err(x.grad.float().cpu(), xr.grad) <= tol
    a = lay.attention
    for p, r in zip((a.Linear.weight, a.Linear.bias, a.OutProj.weight), pr):
        # gradients that cancel analytically (one attention unit: d/d(b1) is a multiple of sum_p d(logit) = 0) are
        # pure summation noise (the oracle's value is 0 or ~1e-8): such a gradient is compared on the absolute scale of
        # the terms it sums (|d(logit)_p * w2| ~ 0.1 with these inputs); every other gradient at north_star's relative bound
        err = float((p.grad.float().cpu() - r.grad).abs().max())
        assert err <= tol * max(float(r.grad.abs().max()), 0.1)
    assert float(a.OutProj.bias.grad.float().abs().max()) <= tol * float(ga.abs().max()) * B    # analytically zero
    # only the output is used downstream: the attention gradient input is None
    x2 = x0.to(dev).requires_grad_()
    y2, _ = lay(x2)
    (y2.rename(None).float() * go.to(dev)).sum().backward()
    xr2 = x0.float().clone().requires_grad_()
    (O.afm_layer(xr2, *[p.detach() for p in pr])[0] * go).sum().backward()
    assert rel_err(x2.grad.float().cpu(), xr2.grad) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("N,E", [(39, 64), (7, 32), (12, 16)])
def test_pair_bilinear_gemm_route_matches_oracle(dev, dtype, tol, N, E):
    """B >= 256 takes the per-field GEMM route (trs_pair_epilogue_*); same checks as the one-kernel route."""
    from torecsys_amd import functional as F_
    from torecsys_amd.layers import BilinearInteractionLayer, OuterProductNetworkLayer
    B = 300
    g = torch.Generator().manual_seed(N + E)
    x0 = (torch.randn(B, N, E, generator=g) * 0.5).to(dtype)
    assert F_._pair_gemm_route(x0.to(dev))
    for fam, kind in (("opn", "mat"), ("bil", "each")):
        torch.manual_seed(9)
        if fam == "opn":
            lay = OuterProductNetworkLayer(E, N, kind).to(dev).to(dtype)
            params = [lay.kernel]
        else:
            lay = BilinearInteractionLayer(E, N, kind).to(dev).to(dtype)
            params = [lay.bilinear.weight, lay.bilinear.bias]
        x = x0.to(dev).requires_grad_()
        y = lay(x).rename(None)
        xr = x0.float().clone().requires_grad_()
        pr = [p.detach().float().cpu().requires_grad_() for p in params]
        yr = O.outer_product_layer(xr, pr[0], kind) if fam == "opn" else O.bilinear_layer(xr, pr[0], pr[1], kind)
        assert rel_err(y.float().cpu(), yr) <= tol, (fam, kind)
        assert rel_err_rows(y.float().cpu(), yr.detach(), floor_frac=5e-2) <= 2 * tol, (fam, kind, "per sample")
        go = torch.randn(yr.shape, generator=g)
        (y.float() * go.to(dev)).sum().backward()
        (yr * go).sum().backward()
        assert rel_err(x.grad.float().cpu(), xr.grad) <= tol, (fam, kind, "gx")
        assert rel_err_rows(x.grad.float().cpu(), xr.grad, floor_frac=5e-2) <= 2 * tol, (fam, kind, "gx per sample")
        for p, r in zip(params, pr):
            assert rel_err(p.grad.float().cpu(), r.grad) <= tol, (fam, kind, "gparam")


# ---- AFM in training mode at the reference's default configuration: dropout on the scores INSIDE the fused pass ----
from conftest import AFM_DROP_SHAPES  # noqa: E402


@pytest.mark.parametrize("shape", AFM_DROP_SHAPES)
def test_afm_score_dropout_golden(golden, dev, shape):
    """afm_drop.npz: the reference in train() with the masks its two nn.Dropout modules drew.  The kernel gets the score
    mask (trs_afm_fwd_dropout / trs_afm_bwd_dropout); the output dropout stays the module the reference has."""
    from torecsys_amd import functional as F_
    G = golden("afm_drop")
    B, N, E = shape
    t = _tag(shape)
    p = float(G(f"{t}/p")[0])
    scale = 1.0 / (1.0 - p)
    ps = [G(f"{t}/{n}").to(dev).requires_grad_() for n in ("W1", "b1", "W2", "b2")]
    x = G(f"{t}/x").to(dev).requires_grad_()
    keep = G(f"{t}/score_keep").to(dev)
    y0, attn = F_.afm(x, *ps, keep, scale)
    assert rel_err(y0.cpu(), G(f"{t}/out_before_dropout")) <= 1e-5
    assert rel_err(attn.unsqueeze(-1).cpu(), G(f"{t}/attn")) <= 1e-5
    assert bool((attn.detach().cpu()[G(f"{t}/score_keep") == 0] == 0).all())       # dropped scores are exactly 0
    y = y0 * G(f"{t}/out_keep").to(dev).float() * scale
    assert rel_err(y.cpu(), G(f"{t}/out")) <= 1e-5
    ((y * G(f"{t}/gout").to(dev)).sum() + (attn.unsqueeze(-1) * G(f"{t}/gattn").to(dev)).sum()).backward()
    assert rel_err(x.grad.cpu(), G(f"{t}/gx")) <= 1e-5
    for p_, n in zip(ps[:3], ("gW1", "gb1", "gW2")):
        assert rel_err(p_.grad.cpu(), G(f"{t}/{n}")) <= 1e-5, n
    assert float((ps[3].grad.cpu() - G(f"{t}/gb2")).abs().max()) <= 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,E,A,p", [(37, 39, 64, 64, 0.1), (130, 7, 24, 40, 0.3), (64, 10, 32, 32, 0.5),
                                       (20, 39, 64, 128, 0.1), (9, 5, 128, 96, 0.2)])
def test_afm_layer_training_dropout_vs_oracle(dev, dtype, tol, B, N, E, A, p):
    """The LAYER in train() with p > 0 (the reference's default p = 0.1): no PyTorch recompute of the (B,NC2,E)
    products -- the mask the layer draws is reproduced here from the same device generator state and handed to the
    oracle.  MFMA path (bf16, E in {32,64,128}) and generic path (fp32 / E = 24)."""
    from torecsys_amd.layers import AttentionalFactorizationMachineLayer
    g = torch.Generator().manual_seed(B + N + E + A)
    x0 = (torch.randn(B, N, E, generator=g) * 0.7).to(dtype)
    ps = [(torch.randn(A, E, generator=g) / E ** 0.5).to(dtype), (torch.randn(A, generator=g) * 0.1).to(dtype),
          (torch.randn(1, A, generator=g) / A ** 0.5).to(dtype), (torch.randn(1, generator=g) * 0.1).to(dtype)]
    lay = _afm_layer(dev, E, N, A, *ps, dtype=dtype)
    lay.attention.Dropout.p = p            # score dropout on, output dropout off: the output is compared directly
    lay.train()
    P = N * (N - 1) // 2
    torch.manual_seed(1234)
    keep = torch.empty(B, P, dtype=torch.uint8, device=dev).bernoulli_(1.0 - p).cpu()
    torch.manual_seed(1234)
    x = x0.to(dev).requires_grad_()
    y, attn = lay(x)
    assert 0 < int(keep.sum()) < keep.numel()
    assert bool((attn.detach().float().cpu().squeeze(-1)[keep == 0] == 0).all()), "the layer drew a different mask"
    xr = x0.float().clone().requires_grad_()
    pr = [q.float().clone().requires_grad_() for q in ps]
    yr, ar = O.afm_layer(xr, *pr, score_keep=keep, keep_scale=1.0 / (1.0 - p))
    assert rel_err(y.rename(None).float().cpu(), yr) <= tol
    assert rel_err(attn.float().cpu(), ar) <= tol
    go, ga = torch.randn(B, E, generator=g), torch.randn(ar.shape, generator=g)
    ((y.rename(None).float() * go.to(dev)).sum() + (attn.float() * ga.to(dev)).sum()).backward()
    ((yr * go).sum() + (ar * ga).sum()).backward()
    assert rel_err(x.grad.float().cpu(), xr.grad) <= tol
    a = lay.attention
    for q, r in zip((a.Linear.weight, a.Linear.bias, a.OutProj.weight), pr):
        err = float((q.grad.float().cpu() - r.grad).abs().max())
        assert err <= tol * max(float(r.grad.abs().max()), 1e-2)
    lay.eval()                              # eval: dropout is the identity, scores sum to one
    _, attn_e = lay(x0.to(dev))
    assert float((attn_e.float().sum(1) - 1).abs().max()) <= (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,E", [(33, 6, 64), (8, 4, 24), (129, 12, 16)])
def test_bilinear_submodules_stand_alone_forward(dev, dtype, tol, B, N, E):
    """FieldAllTypeBilinear / FieldEachTypeBilinear called directly with gathered (B,P,E) operands, as the reference's
    BilinearInteractionLayer.forward calls them (bilinear_interaction.py:72-76, 144-149): library GEMM + the HIP product /
    bias pass, gradients for both operands and the parameters; off-GPU they raise like every module here."""
    from torecsys_amd.layers import FieldAllTypeBilinear, FieldEachTypeBilinear
    g = torch.Generator().manual_seed(B + N + E)
    r, c = O.pair_indices(N)
    P = len(r)
    x0 = (torch.randn(B, N, E, generator=g) * 0.5).to(dtype)
    for kind in ("all", "each"):
        torch.manual_seed(3)
        lay = (FieldAllTypeBilinear(E, E) if kind == "all" else FieldEachTypeBilinear(P, E, E))
        with pytest.raises(RuntimeError, match="no CPU path"):
            lay(x0[:, r].float(), x0[:, c].float())
        lay = lay.to(dev).to(dtype)
        a = x0[:, r].contiguous().to(dev).requires_grad_()
        b = x0[:, c].contiguous().to(dev).requires_grad_()
        y = lay(a, b)
        ar, br = x0[:, r].float().clone().requires_grad_(), x0[:, c].float().clone().requires_grad_()
        Wr = lay.weight.detach().float().cpu().requires_grad_()
        br_ = lay.bias.detach().float().cpu().requires_grad_()
        yr = (torch.matmul(ar, Wr) if kind == "all" else torch.einsum("bpe,peh->bph", ar, Wr)) * br + br_
        assert rel_err(y.float().cpu(), yr) <= tol, kind
        go = torch.randn(yr.shape, generator=g)
        (y.float() * go.to(dev)).sum().backward()
        (yr * go).sum().backward()
        for got, ref, n in ((a.grad, ar.grad, "g1"), (b.grad, br.grad, "g2"), (lay.weight.grad, Wr.grad, "gW"),
                            (lay.bias.grad, br_.grad, "gb")):
            assert rel_err(got.float().cpu(), ref) <= tol, (kind, n)


def test_field_each_type_bilinear_rectangular(dev):
    """in1_features != in2_features (bilinear_interaction.py:143-148: input1 (..., P, in1), input2 (..., P, in2), weight
    (P, in1, in2)) -- round 4 rejected it with a shape check the reference does not have."""
    from torecsys_amd.layers import FieldEachTypeBilinear
    g = torch.Generator().manual_seed(12)
    B, P, E1, E2 = 17, 6, 24, 40
    torch.manual_seed(4)
    lay = FieldEachTypeBilinear(P, E1, E2).to(dev)
    a = torch.randn(B, P, E1, generator=g).to(dev).requires_grad_()
    b = torch.randn(B, P, E2, generator=g).to(dev).requires_grad_()
    y = lay(a, b)
    assert y.shape == (B, P, E2)
    ar, br = a.detach().cpu().clone().requires_grad_(), b.detach().cpu().clone().requires_grad_()
    Wr, cr = lay.weight.detach().cpu().requires_grad_(), lay.bias.detach().cpu().requires_grad_()
    yr = torch.einsum("bpe,peh->bph", ar, Wr) * br + cr
    assert rel_err(y.detach().cpu(), yr.detach()) <= 1e-5
    go = torch.randn(yr.shape, generator=g)
    (y * go.to(dev)).sum().backward()
    (yr * go).sum().backward()
    for got, ref in ((a.grad, ar.grad), (b.grad, br.grad), (lay.weight.grad, Wr.grad), (lay.bias.grad, cr.grad)):
        assert rel_err(got.cpu(), ref) <= 1e-5
    with pytest.raises(ValueError):
        lay(a, b[:, :5])
