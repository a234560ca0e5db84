"""Parity at the BASELINE.json size (B = 65 536, N = 39, E = 64, bf16) through properties that do not need the
oracle to process the full batch: every interaction layer works sample by sample (CIN's BatchNorm in eval mode), so
the kernels run on the whole batch and a random sample of rows is compared with the CPU oracle evaluated on just those
rows; the parameter gradients (sums over the batch) are checked through linearity: the gradient of a loss that
weights only the sampled rows equals the oracle's gradient on those rows."""
import pytest
import torch

from conftest import rel_err
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

B, N, E = 65536, 39, 64
S = 48                       # sampled rows
TOL = 2e-2                   # bf16 (north-star tolerance for the interaction sums: 1e-2; gradients 2e-2)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def block(dev):
    g = torch.Generator().manual_seed(4242)
    x = (torch.randn(B, N, E, generator=g) * 0.5).bfloat16()
    rows = torch.randperm(B, generator=g)[:S].sort().values
    return x, rows, g


def _sampled_loss(y, rows, gsel):
    """sum over the sampled rows only of y * gsel -- every other row gets a zero gradient"""
    return (y.rename(None)[rows.to(y.device)].float() * gsel.to(y.device)).sum()


def test_inner_product_full_size(dev, block):
    from torecsys_amd.layers import InnerProductNetworkLayer
    x0, rows, g = block
    x = x0.to(dev).requires_grad_()
    y = InnerProductNetworkLayer(N)(x)
    xr = x0[rows].float().requires_grad_()
    yr = O.inner_product_layer(xr)
    assert y.shape == (B, N * (N - 1) // 2)
    assert rel_err(y.rename(None)[rows.to(dev)].float().cpu(), yr) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= TOL
    mask = torch.ones(B, dtype=torch.bool); mask[rows] = False
    assert float(x.grad[mask.to(dev)].float().abs().max()) == 0.0          # untouched rows: exactly zero gradient


def test_cross_network_full_size(dev, block):
    from torecsys_amd.layers import CrossNetworkLayer
    x0, rows, g = block
    torch.manual_seed(11)
    lay = CrossNetworkLayer(inputs_size=E, num_layers=6).to(dev).bfloat16()
    x = x0.to(dev).requires_grad_()
    y = lay(x)
    Ws = [l.weight.detach().float().cpu().requires_grad_() for l in lay.model]
    bs = [l.bias.detach().float().cpu().requires_grad_() for l in lay.model]
    xr = x0[rows].float().requires_grad_()
    yr = O.cross_network(xr, Ws, bs)
    assert rel_err(y.rename(None)[rows.to(dev)].float().cpu(), yr) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= TOL
    for l, w, b in zip(lay.model, Ws, bs):
        assert rel_err(l.weight.grad.float().cpu(), w.grad) <= 3e-2
        assert rel_err(l.bias.grad.float().cpu(), b.grad) <= 3e-2


def test_cin_full_size_eval_batchnorm(dev, block):
    """xDeepFM's CIN [128,128,128] at the full batch, BatchNorm in eval mode (running statistics): rows are
    independent, so sampled rows must match the oracle fed only those rows."""
    from torecsys_amd.layers import CompressInteractionNetworkLayer
    x0, rows, g = block
    torch.manual_seed(12)
    lay = CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=1, layer_sizes=[128, 128, 128])
    for seq in lay.model:
        seq.Batchnorm.running_mean.normal_(0.0, 0.05)
        seq.Batchnorm.running_var.uniform_(0.5, 1.5)
    lay = lay.to(dev).bfloat16().eval()
    x = x0.to(dev).requires_grad_()
    y = lay(x)
    f32 = lambda t: t.detach().float().cpu()
    xr = x0[rows].float().requires_grad_()
    yr = O.cin_layer(xr, [f32(seq.Conv1d.weight) for seq in lay.model], [f32(seq.Conv1d.bias) for seq in lay.model],
                     f32(lay.fc.weight), f32(lay.fc.bias),
                     bn_weights=[f32(seq.Batchnorm.weight) for seq in lay.model],
                     bn_biases=[f32(seq.Batchnorm.bias) for seq in lay.model],
                     bn_running_means=[f32(seq.Batchnorm.running_mean) for seq in lay.model],
                     bn_running_vars=[f32(seq.Batchnorm.running_var) for seq in lay.model], training=False)
    assert y.shape == (B, 1)
    assert rel_err(y.rename(None)[rows.to(dev)].float().cpu(), yr) <= 3e-2
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    # three stacked bf16 layers with ReLU masks: a rounding that flips a mask moves the input gradient by a whole term
    # (tests/test_gpu_layers.py pins the MFMA path to the generic bf16 path for the same reason); max-norm bound 0.15
    assert rel_err(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= 0.15


def test_pair_layers_full_size(dev, block):
    """OPN 'mat' (per-field GEMM route), Bilinear 'all', AFM (MFMA) on the full batch, sampled rows vs the oracle."""
    from torecsys_amd.layers import (AttentionalFactorizationMachineLayer, BilinearInteractionLayer,
                                     OuterProductNetworkLayer)
    x0, rows, g = block
    rd = rows.to(dev)
    torch.manual_seed(13)
    opn = OuterProductNetworkLayer(E, N, "mat").to(dev).bfloat16()
    x = x0.to(dev).requires_grad_()
    y = opn(x)
    xr = x0[rows].float().requires_grad_()
    kr = opn.kernel.detach().float().cpu().requires_grad_()
    yr = O.outer_product_layer(xr, kr, "mat")
    assert rel_err(y.rename(None)[rd].float().cpu(), yr) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rd].float().cpu(), xr.grad) <= TOL
    assert rel_err(opn.kernel.grad.float().cpu(), kr.grad) <= 3e-2
    del y, x, opn
    torch.cuda.empty_cache()

    bil = BilinearInteractionLayer(E, N, "all").to(dev).bfloat16()
    x = x0.to(dev).requires_grad_()
    y = bil(x)
    xr = x0[rows].float().requires_grad_()
    Wr = bil.bilinear.weight.detach().float().cpu().requires_grad_()
    br = bil.bilinear.bias.detach().float().cpu().requires_grad_()
    yr = O.bilinear_layer(xr, Wr, br, "all")
    assert rel_err(y.rename(None)[rd].float().cpu(), yr) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rd].float().cpu(), xr.grad) <= TOL
    assert rel_err(bil.bilinear.weight.grad.float().cpu(), Wr.grad) <= 3e-2
    assert rel_err(bil.bilinear.bias.grad.float().cpu(), br.grad) <= 3e-2
    del y, x, bil
    torch.cuda.empty_cache()

    afm = AttentionalFactorizationMachineLayer(E, N, 64, 0.0).to(dev).bfloat16()
    x = x0.to(dev).requires_grad_()
    y, attn = afm(x)
    a = afm.attention
    ps = [p.detach().float().cpu().requires_grad_() for p in (a.Linear.weight, a.Linear.bias, a.OutProj.weight, a.OutProj.bias)]
    xr = x0[rows].float().requires_grad_()
    yr, ar = O.afm_layer(xr, *ps)
    assert rel_err(y.rename(None)[rd].float().cpu(), yr) <= TOL
    assert rel_err(attn[rd].float().cpu(), ar) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rd].float().cpu(), xr.grad) <= 3e-2
    assert rel_err(a.Linear.weight.grad.float().cpu(), ps[0].grad) <= 3e-2
    assert rel_err(a.OutProj.weight.grad.float().cpu(), ps[2].grad) <= 3e-2
