"""Seeded shape fuzz: every op of the path on ragged / degenerate shapes (B=1, N=1, N=2, E=1, odd E, E not a
multiple of the vector width, single-row tables, repeated indices) in fp32 and bf16 against the CPU oracle."""
import random

import pytest
import torch

from conftest import rel_err
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1), (1, 2, 8), (3, 1, 16), (2, 2, 1), (5, 3, 3), (7, 4, 12), (1, 39, 64), (9, 5, 20), (4, 17, 32),
          (6, 2, 128), (11, 7, 24), (2, 40, 16), (3, 64, 32), (13, 6, 36)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _mk(B, N, E, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    rnd = random.Random(seed)
    fs = [rnd.choice([1, 1, 2, 3, 5, 9]) for _ in range(N)]
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1)
    w = torch.randn(sum(fs), E, generator=g).to(dtype)
    x = (0.7 * torch.randn(B, N, E, generator=g)).to(dtype)
    return g, fs, idx, w, x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_lookup_fm_scatter(dev, dtype, shape):
    from torecsys_amd import functional as F_
    B, N, E = shape
    g, fs, idx, w, _ = _mk(B, N, E, dtype, 11 + B + 3 * N + 7 * E)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    off = O.field_offsets(fs)
    wr = w.float().clone().requires_grad_()
    er = O.multi_indices_embedding(wr, idx, off)
    fr = O.fm_layer(er)
    ge = torch.randn(B, N, E, generator=g).to(dtype)
    gf = torch.randn(B, E, generator=g).to(dtype)
    ((er * ge.float()).sum() + (fr * gf.float()).sum()).backward()
    wd = w.to(dev).requires_grad_()
    emb, fm, _ = F_.embed_fm(wd, idx.to(dev).to(torch.int32 if B % 2 else torch.int64), off.to(dev))
    assert torch.equal(emb.cpu(), er.detach().to(dtype))
    scale = float(fr.detach().abs().max().clamp_min(1.0))
    assert float((fm.float().cpu() - fr.detach()).abs().max()) <= tol * scale * 4
    ((emb.float() * ge.to(dev).float()).sum() + (fm.float() * gf.to(dev).float()).sum()).backward()
    assert rel_err(wd.grad.float().cpu(), wr.grad) <= tol
    # plain gather + separate FM layer
    wd2 = w.to(dev).requires_grad_()
    e2 = F_.gather_rows(wd2, idx.to(dev), off.to(dev))
    f2 = F_.fm_layer(e2)
    assert torch.equal(e2.detach().cpu(), er.detach().to(dtype))
    ((e2.float() * ge.to(dev).float()).sum() + (f2.float() * gf.to(dev).float()).sum()).backward()
    assert rel_err(wd2.grad.float().cpu(), wr.grad) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_pair_layers(dev, dtype, shape):
    from torecsys_amd import functional as F_
    B, N, E = shape
    g, fs, idx, w, x = _mk(B, N, E, dtype, 101 + B + 3 * N + 7 * E)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    P = N * (N - 1) // 2
    # IPN
    xd, xr = x.to(dev).requires_grad_(), x.float().clone().requires_grad_()
    y = F_.pair_dot(xd)
    assert y.shape == (B, P)
    if P:
        yr = O.inner_product_layer(xr)
        go = torch.randn(B, P, generator=g).to(dtype)
        assert rel_err(y.float().cpu(), yr.detach()) <= tol
        (y.float() * go.to(dev).float()).sum().backward()
        (yr * go.float()).sum().backward()
        assert rel_err(xd.grad.float().cpu(), xr.grad) <= tol
    # FFM on a materialised block and fused from the tables
    if N <= 17:
        xf = (0.7 * torch.randn(B, N * N, E, generator=g)).to(dtype)
        xfd, xfr = xf.to(dev).requires_grad_(), xf.float().clone().requires_grad_()
        yf = F_.ffm_layer(xfd, N)
        assert yf.shape == (B, P, E)
        if P:
            yfr = O.ffm_layer(xfr, N)
            assert rel_err(yf.float().cpu(), yfr.detach()) <= tol
            gf = torch.randn(B, P, E, generator=g).to(dtype)
            (yf.float() * gf.to(dev).float()).sum().backward()
            (yfr * gf.float()).sum().backward()
            assert rel_err(xfd.grad.float().cpu(), xfr.grad) <= tol
            ws = [torch.randn(sum(fs), E, generator=g).to(dtype) for _ in range(N)]
            wsd = [t.to(dev).requires_grad_() for t in ws]
            wsr = [t.float().clone().requires_grad_() for t in ws]
            off = O.field_offsets(fs)
            yq = F_.ffm_fused(wsd, idx.to(dev), off.to(dev))
            yqr = O.ffm_layer(O.multi_indices_field_aware_embedding(wsr, idx, off), N)
            assert rel_err(yq.float().cpu(), yqr.detach()) <= tol
            (yq.float() * gf.to(dev).float()).sum().backward()
            (yqr * gf.float()).sum().backward()
            for a, b in zip(wsd, wsr):
                assert float((a.grad.float().cpu() - b.grad).abs().max()) <= tol * 4 * max(1.0, float(b.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_cross_and_cin_contract(dev, dtype, shape):
    from torecsys_amd import functional as F_
    B, N, E = shape
    g, fs, idx, w, x = _mk(B, N, E, dtype, 201 + B + 3 * N + 7 * E)
    tol = 1e-5 if dtype == torch.float32 else 1e-2          # north_star's bounds, no multipliers (measured max 7.3e-3)
    L = 1 + (B + N) % 4
    W = (torch.randn(L, E, E, generator=g) / max(1.0, E ** 0.5)).to(dtype)
    b = (0.1 * torch.randn(L, E, generator=g)).to(dtype)
    xs = (0.5 * x)
    xd, Wd, bd = xs.to(dev).requires_grad_(), W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    xr, Wr, br = xs.float().clone().requires_grad_(), W.float().clone().requires_grad_(), b.float().clone().requires_grad_()
    y = F_.cross_network(xd, Wd, bd)
    yr = O.cross_network(xr, list(Wr), list(br))
    assert rel_err(y.float().cpu(), yr.detach()) <= tol
    go = torch.randn(B, N, E, generator=g).to(dtype)
    (y.float() * go.to(dev).float()).sum().backward()
    (yr * go.float()).sum().backward()
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= tol
    assert rel_err(Wd.grad.float().cpu(), Wr.grad) <= tol
    assert rel_err(bd.grad.float().cpu(), br.grad) <= tol
    # CIN contraction (channels-first generic kernels)
    H, C = 1 + (B * 3) % 5, 2 + N % 4
    xk = (0.7 * torch.randn(B, H, E, generator=g)).to(dtype)
    Wc = (torch.randn(C, N * H, generator=g) / (N * H) ** 0.5).to(dtype)
    bc = (0.1 * torch.randn(C, generator=g)).to(dtype)
    gy = torch.randn(B, C, E, generator=g).to(dtype)
    ts = [t.float().clone().requires_grad_() for t in (x, xk, Wc, bc)]
    z = (ts[0].unsqueeze(2) * ts[1].unsqueeze(1)).reshape(B, N * H, E)
    yc_r = torch.einsum("ck,bke->bce", ts[2], z) + ts[3].view(1, C, 1)
    (yc_r * gy.float()).sum().backward()
    td = [t.to(dev).requires_grad_() for t in (x, xk, Wc, bc)]
    yc = F_.cin_contract(*td)
    assert rel_err(yc.float().cpu(), yc_r.detach()) <= tol
    (yc.float() * gy.to(dev).float()).sum().backward()
    for a, r in zip(td, ts):
        assert rel_err(a.grad.float().cpu(), r.grad) <= tol


@pytest.mark.parametrize("E", [32, 64])
@pytest.mark.parametrize("N", [2, 5, 39])
def test_pair_bilinear_mfma_batch_tails(dev, N, E):
    """The matrix-core per-pair bilinear kernels walk 16-sample tiles with loads two tiles ahead and every wave takes an
    equal share of (task, tile) units: batch sizes around every boundary of that scheme (one tile, an odd number of
    tiles, a ragged last tile, fewer units than waves, a wave range that straddles two tasks), OPN 'mat' and Bilinear
    'each', forward and gradients against the same sums in fp32 torch ops on the bf16-rounded operands."""
    from torecsys_amd import functional as F_
    P = N * (N - 1) // 2
    i_idx, j_idx = torch.triu_indices(N, N, 1)
    for B in (16, 17, 31, 32, 33, 47, 48, 49, 65, 100, 257, 1000):
        g = torch.Generator().manual_seed(B * 131 + N * 7 + E)
        x = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16()
        W = (torch.randn(P, E, E, generator=g) / E ** 0.5).bfloat16()
        bias = (0.1 * torch.randn(P, E, generator=g)).bfloat16()
        for mode in (0, 1):
            xd = x.to(dev).requires_grad_()
            Wd = W.to(dev).requires_grad_()
            bd = bias.to(dev).requires_grad_() if mode == 1 else None
            assert F_._pair_mfma_fwd_ok(xd)
            y = F_._PairBilinear.apply(xd, Wd, bd, mode)
            xr = x.float().requires_grad_()
            Wr = W.float().requires_grad_()
            T = torch.einsum("bpe,peh->bph", xr[:, i_idx], Wr)
            yr = (T * xr[:, j_idx]).sum(-1) if mode == 0 else T * xr[:, j_idx] + bias.float()
            assert rel_err(y.float().cpu(), yr) <= 1e-2, (B, mode, "forward")
            go = torch.randn(yr.shape, generator=g)
            (y.float() * go.to(dev)).sum().backward()
            (yr * go).sum().backward()
            assert rel_err(xd.grad.float().cpu(), xr.grad) <= 1e-2, (B, mode, "gx")
            assert rel_err(Wd.grad.float().cpu(), Wr.grad) <= 1e-2, (B, mode, "gW")
