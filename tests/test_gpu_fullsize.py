"""Parity at the BASELINE.json size (B = 65 536, N = 39, E = 64, bf16) through properties that do not need the
oracle to process the full batch: every interaction layer works sample by sample (CIN's BatchNorm in eval mode, or in
train mode once the batch statistics are given), so the kernels run on the whole batch and a random sample of rows is
compared with the CPU oracle evaluated on just those rows; the parameter gradients (sums over the batch) are checked
through linearity: the gradient of a loss that weights only the sampled rows equals the oracle's gradient on those rows.

Bounds: north_star's 1e-2 (bf16) everywhere, in the max norm AND per sampled row (``rel_err_rows``: every row is
normalised by its own largest reference value, so a row of small values cannot hide behind a large one elsewhere).
Two documented exceptions, both derived where they are used: a sum with cancellation is bounded relative to the
magnitude of its terms (``sum_err``), and gradients behind ReLU masks are compared under the kernel's own masks."""
import pytest
import torch

from conftest import rel_err, rel_err_rows, sum_err
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

B, N, E = 65536, 39, 64
S = 48                       # sampled rows
TOL = 1e-2                   # north_star: 1e-2 relative for bf16 interaction sums


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
    assert rel_err_rows(y.rename(None)[rows.to(dev)].float().cpu(), yr) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= TOL
    assert rel_err_rows(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= TOL
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
    assert rel_err_rows(y.rename(None)[rows.to(dev)].float().cpu(), yr) <= TOL
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    (yr * gs).sum().backward()
    assert rel_err(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= TOL
    assert rel_err_rows(x.grad[rows.to(dev)].float().cpu(), xr.grad) <= TOL
    for l, w, b in zip(lay.model, Ws, bs):
        assert rel_err(l.weight.grad.float().cpu(), w.grad) <= TOL
        assert rel_err(l.bias.grad.float().cpu(), b.grad) <= TOL


def _cin_params(lay):
    f32 = lambda t: t.detach().float().cpu()
    return dict(conv_weights=[f32(seq.Conv1d.weight).requires_grad_() for seq in lay.model],
                conv_biases=[f32(seq.Conv1d.bias) for seq in lay.model],
                fc_weight=f32(lay.fc.weight).requires_grad_(), fc_bias=f32(lay.fc.bias),
                bn_weights=[f32(seq.Batchnorm.weight) for seq in lay.model],
                bn_biases=[f32(seq.Batchnorm.bias) for seq in lay.model])


def _affine_of(yT, bn, mean, var):
    scale = bn.weight.double() / torch.sqrt(var + bn.eps)
    return scale, bn.bias.double() - mean * scale


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_cin_full_size(dev, block, mode):
    """xDeepFM's CIN [128,128,128] at the full batch -- what ``bench.py --model xdeepfm`` times.
    eval: BatchNorm on running statistics, rows independent: sampled rows against the oracle fed only those rows,
    gradients under the kernel's own ReLU masks (see tests/test_gpu_cin_parity.py for why).
    train: BatchNorm on BATCH statistics.  Given the statistics the rows are independent again: each layer's batch
    mean / variance is reduced in float64 on the device from the layer's own contraction output and handed to the
    oracle as its BatchNorm statistics; the sampled rows' forward must then match.  (The train-mode backward couples
    all rows through the statistics; it is pinned kernel by kernel at this batch size in test_cin_pieces_full_size.)"""
    from test_gpu_cin_parity import _GlueRecorder
    from torecsys_amd import functional as F_
    from torecsys_amd.layers import CompressInteractionNetworkLayer
    x0, rows, g = block
    rd = rows.to(dev)
    torch.manual_seed(12)
    lay = CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=1, layer_sizes=[128, 128, 128])
    for seq in lay.model:
        seq.Batchnorm.running_mean.normal_(0.0, 0.05)
        seq.Batchnorm.running_var.uniform_(0.5, 1.5)
        seq.Batchnorm.weight.data.uniform_(0.5, 1.5)
        seq.Batchnorm.bias.data.normal_(0.0, 0.2)
    lay = lay.to(dev).bfloat16()
    lay.train(mode == "train")
    x = x0.to(dev).requires_grad_()
    with _GlueRecorder(F_) as rec:
        y = lay(x)
    assert y.shape == (B, 1) and len(rec.seen) == 3
    P = _cin_params(lay)
    stats, masks = [], []
    for (yT, bn, D, Hs) in rec.seen:
        if mode == "train":
            y64 = yT.double()
            mean, var = y64.mean(dim=(0, 1)), y64.var(dim=(0, 1), unbiased=False)
            del y64
        else:
            mean, var = bn.running_mean.double(), bn.running_var.double()
        scale, shift = _affine_of(yT, bn, mean, var)
        z = yT[rd].float() * scale.float() + shift.float()                         # sampled rows, (S,E,C)
        masks.append((z > 0).float().transpose(1, 2).contiguous().cpu())
        stats.append((mean.float().cpu(), var.float().cpu()))
    xr = x0[rows].float().requires_grad_()
    yr, inter, pooled = O.cin_layer(xr, **P, bn_running_means=[m.clone() for m, _ in stats],
                                    bn_running_vars=[v.clone() for _, v in stats], training=False,
                                    return_intermediates=True)
    terms = pooled.detach().abs() @ P["fc_weight"].detach().abs().t() + P["fc_bias"].abs()
    assert sum_err(y.rename(None)[rd].float().cpu(), yr.detach(), terms) <= TOL
    if mode == "train":
        return
    gs = torch.randn(yr.shape, generator=g)
    _sampled_loss(y, rows, gs).backward()
    acts = [(lambda t, m=m: t * m) for m in masks]
    xm = x0[rows].float().requires_grad_()
    ym = O.cin_layer(xm, **P, bn_running_means=[m.clone() for m, _ in stats],
                     bn_running_vars=[v.clone() for _, v in stats], training=False, activation=acts)
    (ym * gs).sum().backward()
    assert rel_err(x.grad[rd].float().cpu(), xm.grad) <= TOL
    assert rel_err_rows(x.grad[rd].float().cpu(), xm.grad, floor_frac=5e-2) <= 2 * TOL
    for k, seq in enumerate(lay.model):
        assert rel_err(seq.Conv1d.weight.grad.float().cpu(), P["conv_weights"][k].grad) <= TOL, k
    assert rel_err(lay.fc.weight.grad.float().cpu(), P["fc_weight"].grad) <= TOL
    mask = torch.ones(B, dtype=torch.bool); mask[rows] = False
    assert float(x.grad[mask.to(dev)].float().abs().max()) == 0.0


def test_cin_pieces_full_size(dev, block):
    """The kernels of one train-mode CIN layer, each alone, at B = 65 536 (H = 128, C = 256):
    contraction forward / data gradient / weight gradient on sampled rows against ``oracle.cin_contraction``
    (rows independent; weight gradient through a loss on the sampled rows), and the train-mode glue
    (BatchNorm1d batch statistics + ReLU + chunk + pooled sum) forward and backward on the WHOLE batch against the
    same sequence in fp32 torch ops on the device (F.batch_norm, relu, chunk, sum)."""
    from torecsys_amd import functional as F_
    x0, rows, g = block
    rd = rows.to(dev)
    H, C = 128, 256
    ld0 = 64
    x0T = torch.zeros(B, E, ld0, dtype=torch.bfloat16, device=dev)
    x0T[:, :, :N] = x0.to(dev).transpose(1, 2)
    x0T.requires_grad_()
    xk = (0.5 * torch.randn(B, H, E, generator=g)).abs_().bfloat16()
    xkT = xk.to(dev).transpose(1, 2).contiguous().requires_grad_()
    W = (torch.randn(C, N * H, generator=g) / (N * H) ** 0.5).bfloat16()
    bias = (0.1 * torch.randn(C, generator=g)).bfloat16()
    Wd, bd = W.to(dev).requires_grad_(), bias.to(dev).requires_grad_()
    yT = F_.cin_contract_cl(x0T, xkT, Wd, bd, N, H)
    gsel = torch.zeros(B, E, C, dtype=torch.bfloat16, device=dev)
    gs = torch.randn(S, C, E, generator=g).bfloat16()
    gsel[rd] = gs.to(dev).transpose(1, 2)
    d0, dk, dW = torch.autograd.grad(yT, (x0T, xkT, Wd), gsel)
    a = x0[rows].float().permute(0, 2, 1).requires_grad_()
    k = xk[rows].float().permute(0, 2, 1).requires_grad_()
    Wo = W.float().reshape(C, N * H, 1).requires_grad_()
    yo = O.cin_contraction(a, k, Wo, bias.float())
    (yo * gs.float()).sum().backward()
    ys = yT.detach()[rd].transpose(1, 2).float().cpu()
    assert rel_err(ys, yo.detach()) <= TOL and rel_err_rows(ys, yo.detach()) <= TOL
    assert rel_err_rows(d0[rd][:, :, :N].float().cpu(), a.grad) <= TOL
    assert rel_err_rows(dk[rd].float().cpu(), k.grad) <= TOL
    assert rel_err(dW.float().cpu(), Wo.grad.reshape(C, N * H)) <= TOL
    assert rel_err_rows(dW.float().cpu(), Wo.grad.reshape(C, N * H)) <= TOL
    del d0, dk, gsel
    # ---- glue, train mode, whole batch
    D = Hs = C // 2
    bn = torch.nn.BatchNorm1d(C).to(dev).bfloat16().train()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0.0, 0.2)
    ya = yT.detach().clone().requires_grad_()
    hidden, pooled = F_.cin_glue(ya, bn, D, Hs)
    gh = torch.randn(hidden.shape, device=dev, dtype=torch.bfloat16)
    gp = torch.randn(pooled.shape, device=dev, dtype=torch.bfloat16)
    torch.autograd.backward((hidden, pooled), (gh, gp))
    yb = yT.detach().float().reshape(B * E, C).requires_grad_()
    gam, bet = bn.weight.detach().float().requires_grad_(), bn.bias.detach().float().requires_grad_()
    z = torch.relu(torch.nn.functional.batch_norm(yb, None, None, gam, bet, True, 0.1, bn.eps)).reshape(B, E, C)
    ref_hidden, ref_pooled = z[:, :, Hs:], z[:, :, :D].sum(dim=1)
    torch.autograd.backward((ref_hidden, ref_pooled), (gh.float(), gp.float()))
    assert rel_err(hidden.float(), ref_hidden.detach()) <= TOL
    assert rel_err(pooled.float(), ref_pooled.detach()) <= TOL
    assert rel_err_rows(pooled.float(), ref_pooled.detach()) <= TOL
    assert rel_err(ya.grad.float(), yb.grad.reshape(B, E, C)) <= TOL
    assert rel_err(bn.weight.grad.float(), gam.grad) <= TOL
    assert rel_err(bn.bias.grad.float(), bet.grad) <= TOL


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
    assert rel_err(opn.kernel.grad.float().cpu(), kr.grad) <= TOL
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
    assert rel_err(bil.bilinear.weight.grad.float().cpu(), Wr.grad) <= TOL
    assert rel_err(bil.bilinear.bias.grad.float().cpu(), br.grad) <= TOL
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
    assert rel_err(x.grad[rd].float().cpu(), xr.grad) <= TOL
    assert rel_err(a.Linear.weight.grad.float().cpu(), ps[0].grad) <= TOL
    assert rel_err(a.OutProj.weight.grad.float().cpu(), ps[2].grad) <= TOL


def test_ffm_full_size(dev):
    """F2 / I3 at the BASELINE shape (12.8 GB of field-aware rows in, 6.2 GB of pair products out; 5 GB of
    tables): the fused lookup + FFM (trs_ffm_fused_fwd -- (B, N*N, E) is never formed) on the whole batch against the
    oracle's field-aware lookup + FFM on 48 sampled rows.  The products are single bf16 multiplications of table rows:
    BIT-EXACT (field_aware_factorization_machine.py:69-87, multi_indices_field_aware_emb.py:102-105).  Element counts
    pass 2^32 here (B*N*N*E = 6.4e9), which is what the small-shape tests cannot reach.  Backward: a loss that weights
    only the sampled rows must reproduce the oracle's table gradients on the rows those samples touch and leave every
    other row exactly zero (checked on two of the 39 tables)."""
    from torecsys_amd import functional as F_
    V = 1_000_000
    per = V // N
    fs = [per] * (N - 1) + [V - per * (N - 1)]
    off = O.field_offsets(fs)
    g = torch.Generator().manual_seed(99)
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1)
    rows = torch.randperm(B, generator=g)[:S].sort().values
    gd = torch.Generator(device=dev).manual_seed(5)
    tabs = [((torch.rand(V, E, generator=gd, device=dev) - 0.5)).bfloat16().requires_grad_() for _ in range(N)]
    y = F_.ffm_fused(tabs, idx.to(dev), off.to(dev))
    P = N * (N - 1) // 2
    assert y.shape == (B, P, E)
    # oracle on the sampled rows: only the table rows those samples read are needed on the host
    gsel = (idx[rows] + off.view(1, -1)).reshape(-1)                      # global row ids, (S*N)
    uniq, inv = torch.unique(gsel, return_inverse=True)
    small = [t.detach()[uniq.to(dev)].float().cpu().requires_grad_() for t in tabs]
    xr = O.multi_indices_field_aware_embedding(small, inv.view(S, N), torch.zeros(N, dtype=torch.int64))
    yr = O.ffm_layer(xr, N)
    got = y[rows.to(dev)].float().cpu()
    assert torch.equal(got, yr.detach().bfloat16().float())
    gs = torch.randn(S, P, E, generator=g)
    _sampled_loss(y, rows, gs.bfloat16().float()).backward()
    (yr * gs.bfloat16().float()).sum().backward()
    for i in (0, N - 1):
        gi = tabs[i].grad
        assert rel_err(gi[uniq.to(dev)].float().cpu(), small[i].grad) <= TOL
        mask = torch.ones(V, dtype=torch.bool); mask[uniq] = False
        assert float(gi[mask.to(dev)].float().abs().max()) == 0.0
