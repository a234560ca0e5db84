"""GPU parity of the SURVEY.md 8f N3 layers (OuterProductNetwork, BilinearInteraction, AFM) through the C ABI:
fp32 against golden vectors captured from the reference (tests/golden/pairs.npz), bf16 and other shapes against the
CPU oracle on the same inputs."""
import pytest
import torch

from conftest import PAIR_SHAPES, pair_heavy_ok, rel_err
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


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,N,E", [(37, 39, 64), (5, 2, 16), (130, 7, 24), (64, 5, 128), (1, 3, 8)])
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
        go = torch.randn(yr.shape, generator=g)
        (y.float() * go.to(dev)).sum().backward()
        (yr * go).sum().backward()
        assert rel_err(x.grad.float().cpu(), xr.grad) <= tol * 2, (fam, kind, "gx")
        for p, r in zip(params, pr):
            assert rel_err(p.grad.float().cpu(), r.grad) <= tol * 2, (fam, kind, "gparam")
