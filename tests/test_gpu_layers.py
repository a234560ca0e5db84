"""GPU parity for the interaction layers (F2-F5) through the C ABI against golden vectors from the
reference: fp32 within 1e-5 relative (FFM products exact), bf16 within 1e-2 of the fp32 oracle on
bf16-rounded inputs."""
import pytest
import torch

from conftest import CIN_CASES, LAYER_SHAPES, rel_err_both as rel_err      # rel_err here = max norm AND per-row norm (conftest.rel_err_both)
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
TOL32 = 1e-5
TOLBF = 1e-2


def _tag(s):
    return "%d_%d_%d" % s


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_ipn_golden(golden, dev, shape):
    from torecsys_amd.layers import InnerProductNetworkLayer
    G = golden("layers")
    tag = _tag(shape)
    x = G(f"fm/{tag}/x").to(dev).requires_grad_()
    y = InnerProductNetworkLayer(num_fields=shape[1])(x.refine_names("B", "N", "E"))
    assert y.names == ("B", "O")
    assert rel_err(y.rename(None).cpu(), G(f"ipn/{tag}/out")) <= TOL32
    (y.rename(None) * G(f"ipn/{tag}/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(f"ipn/{tag}/gx")) <= TOL32


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_ffm_golden(golden, dev, shape):
    from torecsys_amd.layers import FFMLayer, FieldAwareFactorizationMachineLayer
    assert FFMLayer is FieldAwareFactorizationMachineLayer
    G = golden("layers")
    tag = _tag(shape)
    x = G(f"ffm/{tag}/x").to(dev).requires_grad_()
    y = FFMLayer(num_fields=shape[1])(x.refine_names("B", "N", "E"))
    assert y.names == ("B", "N", "E")
    assert torch.equal(y.rename(None).cpu(), G(f"ffm/{tag}/out"))
    (y.rename(None) * G(f"ffm/{tag}/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(f"ffm/{tag}/gx")) <= TOL32


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_cross_golden(golden, dev, shape):
    from torecsys_amd.layers import CrossNetworkLayer
    G = golden("layers")
    tag = _tag(shape)
    W, b = G(f"cross/{tag}/W"), G(f"cross/{tag}/b")
    lay = CrossNetworkLayer(inputs_size=shape[2], num_layers=W.shape[0]).to(dev)
    assert list(lay.state_dict().keys()) == [f"model.{l}.{k}" for l in range(W.shape[0]) for k in ("weight", "bias")]
    for l, lin in enumerate(lay.model):
        lin.weight.data.copy_(W[l])
        lin.bias.data.copy_(b[l])
    x = G(f"cross/{tag}/x").to(dev).requires_grad_()
    xin = x.refine_names("B", "N", "E")
    y = lay(xin)
    assert y.names == ("B", "N", "O")
    assert xin.names == ("B", "N", "E")          # caller's tensor keeps its names (not the reference's side effect)
    assert rel_err(y.rename(None).cpu(), G(f"cross/{tag}/out")) <= TOL32
    (y.rename(None) * G(f"cross/{tag}/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(f"cross/{tag}/gx")) <= TOL32
    assert rel_err(torch.stack([l.weight.grad for l in lay.model]).cpu(), G(f"cross/{tag}/gW")) <= TOL32
    assert rel_err(torch.stack([l.bias.grad for l in lay.model]).cpu(), G(f"cross/{tag}/gb")) <= TOL32
    with pytest.raises(RuntimeError):            # the reference raises RuntimeError on 2-D input too
        lay(torch.randn(4, shape[2], device=dev))


def _load_cin(G, name, dev):
    from torecsys_amd.layers import CompressInteractionNetworkLayer
    pre = f"cin/{name}"
    B, N, E, direct, use_bias, use_bn = G(pre + "/cfg").tolist()
    sizes = G(pre + "/layer_sizes").tolist()
    lay = CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=3, layer_sizes=sizes,
                                          is_direct=bool(direct), use_bias=bool(use_bias),
                                          use_batchnorm=bool(use_bn), activation=torch.nn.ReLU())
    for i, seq in enumerate(lay.model):
        seq.Conv1d.weight.data.copy_(G(f"{pre}/conv_w{i}"))
        if use_bias:
            seq.Conv1d.bias.data.copy_(G(f"{pre}/conv_b{i}"))
        if use_bn:
            seq.Batchnorm.weight.data.copy_(G(f"{pre}/bn_w{i}"))
            seq.Batchnorm.bias.data.copy_(G(f"{pre}/bn_b{i}"))
    lay.fc.weight.data.copy_(G(pre + "/fc_w"))
    lay.fc.bias.data.copy_(G(pre + "/fc_b"))
    return lay.to(dev), len(sizes), bool(use_bias), bool(use_bn)


@pytest.mark.parametrize("name", CIN_CASES)
def test_cin_golden(golden, dev, name):
    G = golden("cin")
    pre = f"cin/{name}"
    lay, L, use_bias, use_bn = _load_cin(G, name, dev)
    lay.train()
    x = G(pre + "/x").to(dev).requires_grad_()
    y = lay(x.refine_names("B", "N", "E"))
    assert y.names == ("B", "O")
    assert rel_err(y.rename(None).cpu(), G(pre + "/train_out")) <= 1e-5
    (y.rename(None) * G(pre + "/gout").to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), G(pre + "/train_gx")) <= 1e-5
    for i, seq in enumerate(lay.model):
        assert rel_err(seq.Conv1d.weight.grad.cpu(), G(f"{pre}/train_gconv_w{i}")) <= 1e-5
        if use_bias:
            gb = G(f"{pre}/train_gconv_b{i}")
            # with BatchNorm behind the convolution d/d(conv bias) is analytically ZERO (the batch mean absorbs it): both
            # sides hold fp32 summation noise of sums whose terms are O(1), compared on the absolute scale of the terms
            assert float((seq.Conv1d.bias.grad.cpu() - gb).abs().max()) <= (1e-4 if use_bn else 1e-5) * max(1.0, float(gb.abs().max()))
        if use_bn:
            assert rel_err(seq.Batchnorm.weight.grad.cpu(), G(f"{pre}/train_gbn_w{i}")) <= 1e-5
            assert rel_err(seq.Batchnorm.running_mean.cpu(), G(f"{pre}/run_mean{i}")) <= 1e-5
            assert rel_err(seq.Batchnorm.running_var.cpu(), G(f"{pre}/run_var{i}")) <= 1e-5
    assert rel_err(lay.fc.weight.grad.cpu(), G(pre + "/train_gfc_w")) <= 1e-5
    lay.eval()
    x2 = G(pre + "/x").to(dev).requires_grad_()
    y2 = lay(x2)
    assert rel_err(y2.rename(None).cpu(), G(pre + "/eval_out")) <= 1e-5
    (y2.rename(None) * G(pre + "/gout").to(dev)).sum().backward()
    assert rel_err(x2.grad.cpu(), G(pre + "/eval_gx")) <= 1e-5


@pytest.mark.parametrize("B,N,E", [(512, 39, 64), (300, 10, 16), (64, 7, 128), (33, 5, 10)])
def test_layers_bf16_vs_oracle(dev, B, N, E):
    """bf16 kernels (fp32 accumulation, one rounding on store) vs the fp32 oracle on bf16-rounded inputs."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B + N + E)
    x = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16()
    xd = x.to(dev).requires_grad_()
    xr = x.float().requires_grad_()
    # IPN
    go = torch.randn(B, N * (N - 1) // 2, generator=g).bfloat16()
    y = F_.pair_dot(xd)
    yr = O.inner_product_layer(xr)
    assert rel_err(y.float().cpu(), yr.detach()) <= TOLBF
    (y.float() * go.to(dev).float()).sum().backward()
    (yr * go.float()).sum().backward()
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= TOLBF
    # FM layer
    xd.grad = None
    xr.grad = None
    gf = torch.randn(B, E, generator=g).bfloat16()
    y = F_.fm_layer(xd)
    yr = O.fm_layer(xr)
    assert rel_err(y.float().cpu(), yr.detach()) <= TOLBF
    (y.float() * gf.to(dev).float()).sum().backward()
    (yr * gf.float()).sum().backward()
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= TOLBF
    # cross, 3 layers
    L = 3
    W = (torch.randn(L, E, E, generator=g) / E ** 0.5).bfloat16()
    b = (0.1 * torch.randn(L, E, generator=g)).bfloat16()
    Wd, bd = W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    Wr, br = W.float().requires_grad_(), b.float().requires_grad_()
    xd.grad = None
    xr.grad = None
    gc = torch.randn(B, N, E, generator=g).bfloat16()
    y = F_.cross_network(xd, Wd, bd)
    yr = O.cross_network(xr, list(Wr), list(br))
    assert rel_err(y.float().cpu(), yr.detach()) <= TOLBF
    (y.float() * gc.to(dev).float()).sum().backward()
    (yr * gc.float()).sum().backward()
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= TOLBF
    assert rel_err(Wd.grad.float().cpu(), Wr.grad) <= TOLBF
    assert rel_err(bd.grad.float().cpu(), br.grad) <= TOLBF


@pytest.mark.parametrize("B,N,E,H,C", [(64, 39, 64, 16, 32), (32, 6, 16, 8, 12), (16, 10, 8, 5, 7)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cin_contract_vs_oracle(dev, dtype, B, N, E, H, C):
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B + N + E + H + C)
    x0 = (0.7 * torch.randn(B, N, E, generator=g)).to(dtype)
    xk = (0.7 * torch.randn(B, H, E, generator=g)).to(dtype)
    W = (torch.randn(C, N * H, generator=g) / (N * H) ** 0.5).to(dtype)
    bias = (0.1 * torch.randn(C, generator=g)).to(dtype)
    gy = torch.randn(B, C, E, generator=g).to(dtype)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    ts = [t.float().clone().requires_grad_() for t in (x0, xk, W, bias)]
    z = (ts[0].unsqueeze(2) * ts[1].unsqueeze(1)).reshape(B, N * H, E)          # (B, N*H, E), index n*H+h
    yr = torch.einsum("ck,bke->bce", ts[2], z) + ts[3].view(1, C, 1)
    (yr * gy.float()).sum().backward()
    td = [t.to(dev).requires_grad_() for t in (x0, xk, W, bias)]
    y = F_.cin_contract(*td)
    assert rel_err(y.float().cpu(), yr.detach()) <= tol
    (y.float() * gy.to(dev).float()).sum().backward()
    for a, r in zip(td, ts):
        assert rel_err(a.grad.float().cpu(), r.grad) <= 2 * tol


@pytest.mark.parametrize("detach", [True, False])
@pytest.mark.parametrize("B,N,E,L", [(37, 5, 32, 2), (129, 39, 64, 6), (64, 3, 96, 3), (50, 7, 128, 4), (20, 4, 128, 6),
                                     (601, 39, 64, 6), (1, 1, 64, 1), (700, 39, 32, 5), (333, 3, 64, 3)])
def test_cross_mfma_bf16_vs_oracle(dev, B, N, E, L, detach):
    """bf16 MFMA cross path (E % 32 == 0; ragged row counts; resident and non-resident weight fragments; row counts
    that give a workgroup of the backward kernel zero, one and several 64-row groups; with and without the reference's
    detached first input)."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B * 3 + N + E + L)
    x = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16()
    W = (torch.randn(L, E, E, generator=g) / E ** 0.5).bfloat16()
    b = (0.1 * torch.randn(L, E, generator=g)).bfloat16()
    gc = torch.randn(B, N, E, generator=g).bfloat16()
    xd, Wd, bd = x.to(dev).requires_grad_(), W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    xr, Wr, br = x.float().requires_grad_(), W.float().requires_grad_(), b.float().requires_grad_()
    y = F_.cross_network(xd, Wd, bd, detach)
    yr = O.cross_network(xr, list(Wr), list(br), detach)
    assert rel_err(y.float().cpu(), yr.detach()) <= TOLBF
    (y.float() * gc.to(dev).float()).sum().backward()
    (yr * gc.float()).sum().backward()
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= TOLBF
    assert rel_err(Wd.grad.float().cpu(), Wr.grad) <= TOLBF
    assert rel_err(bd.grad.float().cpu(), br.grad) <= TOLBF


# (the bf16 matrix-core CIN path is pinned to the oracle kernel by kernel in tests/test_gpu_cin_parity.py)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,E", [(64, 6, 16), (33, 12, 8), (40, 39, 64), (17, 5, 10)])
def test_fused_ffm_equals_two_module_path(dev, dtype, B, N, E):
    """FusedFieldAwareFM (tables -> pair products, nothing materialised) == field-aware lookup + FFM layer, forward
    (exact: one multiply per element) and all N table gradients."""
    from torecsys_amd.fused import FusedFieldAwareFM
    from torecsys_amd.inputs import MultiIndicesFieldAwareEmbedding
    from torecsys_amd.layers import FFMLayer
    g = torch.Generator().manual_seed(B + N + E)
    fs = [3 + (i % 5) for i in range(N)]
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev)
    torch.manual_seed(1)
    fa = MultiIndicesFieldAwareEmbedding(embed_size=E, field_sizes=fs).to(dev).to(dtype)
    fused = FusedFieldAwareFM(embed_size=E, field_sizes=fs).to(dev).to(dtype)
    fused.load_state_dict(fa.state_dict())
    go = torch.randn(B, N * (N - 1) // 2, E, generator=g).to(dtype).to(dev)
    y_ref = FFMLayer(num_fields=N)(fa(idx))
    (y_ref.rename(None).float() * go.float()).sum().backward()
    y = fused(idx)
    assert y.names == ("B", "N", "E")
    assert torch.equal(y.rename(None), y_ref.rename(None))
    (y.rename(None).float() * go.float()).sum().backward()
    tol = TOL32 if dtype == torch.float32 else TOLBF
    for a, b in zip(fused.embeddings, fa.embeddings):
        assert rel_err(a.weight.grad.float().cpu(), b.weight.grad.float().cpu()) <= tol
    # and against the oracle in fp32
    if dtype == torch.float32:
        ws = [e.weight.detach().cpu().clone().requires_grad_() for e in fa.embeddings]
        off = O.field_offsets(fs)
        yo = O.ffm_layer(O.multi_indices_field_aware_embedding(ws, idx.cpu(), off), N)
        assert rel_err(y.rename(None).detach().cpu(), yo.detach()) <= TOL32
        (yo * go.cpu()).sum().backward()
        assert rel_err(fused.embeddings[2].weight.grad.cpu(), ws[2].grad) <= TOL32


@pytest.mark.parametrize("mode", ["train", "eval", "nobn", "direct", "noaffine"])
@pytest.mark.parametrize("B,E,C", [(33, 64, 256), (7, 16, 64), (130, 8, 128), (5, 32, 512)])
def test_cin_glue_matches_aten_sequence(dev, mode, B, E, C):
    """trs_cin_glue_* (BatchNorm1d + ReLU + chunk + sum over E in two passes) against the ATen sequence the layer
    used before: outputs, input gradient, BatchNorm parameter gradients and running statistics."""
    import copy
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B + E + C)
    y0 = (torch.randn(B, E, C, generator=g) * 1.5 + 0.3).bfloat16().to(dev)
    bn = None
    if mode != "nobn":
        bn = torch.nn.BatchNorm1d(C, affine=(mode != "noaffine")).to(dev)
        if bn.affine:
            bn.weight.data.copy_(torch.rand(C, generator=g) + 0.5)
            bn.bias.data.copy_(torch.randn(C, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
        bn = bn.bfloat16()
        if mode == "eval":
            bn.eval()
    ref_bn = copy.deepcopy(bn)
    D, Hs = (C, 0) if mode == "direct" else (C // 2, C // 2)
    assert F_.cin_glue_supported(y0, D, Hs)
    ya = y0.clone().requires_grad_()
    hidden, pooled = F_.cin_glue(ya, bn, D, Hs)
    # shapes with E == 8 * (256 / (C / 8)) also get channels-first copies of the hidden half and of the input gradient
    cf = bool(F_.size_query("trs_cin_glue_cf_supported", E, C))
    assert cf == ((E, C) in ((64, 256), (32, 512)))
    seen_cf = []
    ya.register_hook(lambda g_: seen_cf.append((g_.detach().clone(), getattr(g_, '_trs_cf', None))))
    if cf and C > Hs:
        assert torch.equal(hidden._trs_cf, hidden.detach().transpose(1, 2).contiguous())
    else:
        assert getattr(hidden, '_trs_cf', None) is None
    yb = y0.clone().requires_grad_()
    z = yb.reshape(B * E, C)
    if ref_bn is not None:
        z = ref_bn(z)
    z = torch.relu(z).reshape(B, E, C)
    ref_hidden, ref_pooled = z[:, :, Hs:], z[:, :, :D].sum(dim=1)
    assert hidden.shape == ref_hidden.shape and pooled.shape == ref_pooled.shape
    assert rel_err(hidden.float().cpu(), ref_hidden.float().cpu()) <= 1e-2
    assert rel_err(pooled.float().cpu(), ref_pooled.float().cpu()) <= 1e-2
    gh = torch.randn(hidden.shape, generator=g).bfloat16().to(dev)
    gp = torch.randn(pooled.shape, generator=g).bfloat16().to(dev)
    ((hidden.float() * gh.float()).sum() + (pooled.float() * gp.float()).sum()).backward()
    ((ref_hidden.float() * gh.float()).sum() + (ref_pooled.float() * gp.float()).sum()).backward()
    assert rel_err(ya.grad.float().cpu(), yb.grad.float().cpu()) <= 1e-2
    if cf:
        gy_seen, gy_cf = seen_cf[0]
        assert gy_cf is not None, "the channels-first gradient copy did not ride on the gradient tensor"
        assert torch.equal(gy_cf, gy_seen.transpose(1, 2).contiguous())
    if bn is not None:
        if bn.affine:
            assert rel_err(bn.weight.grad.float().cpu(), ref_bn.weight.grad.float().cpu()) <= 1e-2
            assert rel_err(bn.bias.grad.float().cpu(), ref_bn.bias.grad.float().cpu()) <= 1e-2
        assert rel_err(bn.running_mean.float().cpu(), ref_bn.running_mean.float().cpu()) <= 1e-2
        assert rel_err(bn.running_var.float().cpu(), ref_bn.running_var.float().cpu()) <= 1e-2
        assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)
    # only one of the two outputs used downstream
    yc = y0.clone().requires_grad_()
    h2, p2 = F_.cin_glue(yc, copy.deepcopy(ref_bn), D, Hs)
    (p2.float() * gp.float()).sum().backward()
    assert torch.isfinite(yc.grad.float()).all()
