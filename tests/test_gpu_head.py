"""The scalar head of the CTR models (functional.ctr_logit, csrc/head.hip) and the loss the benchmark is defined on
(functional.bce_with_logits) against the oracle's model compositions (oracle.fm_model / deepfm_model / xdeepfm_model:
models/ctr/factorization_machine.py:55-66, deep_fm.py:75-104, xdeep_fm.py:117-121) and ATen's BCE-with-logits."""
import pytest
import torch

from conftest import rel_err
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,E", [(257, 39, 64), (1, 1, 1), (1000, 10, 16), (33, 5, 10), (4096, 3, 130)])
def test_ctr_logit_vs_oracle_models(dev, dtype, tol, B, N, E):
    """fm + feat (+ deep) (+ bias): forward and every operand's gradient.  bf16: the kernel accumulates in fp32 and rounds
    once, the oracle runs in fp32 on the bf16-rounded operands."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B + N + E)
    fm = torch.randn(B, E, generator=g).to(dtype)
    feat = torch.randn(B, N, 1, generator=g).to(dtype)
    deep = torch.randn(B, 1, generator=g).to(dtype)
    cin = torch.randn(B, 1, generator=g).to(dtype)
    bias = torch.randn(1, generator=g).to(dtype)
    go = torch.randn(B, 1, generator=g)

    def leaves(ts, d):
        return [t.to(d).clone().requires_grad_() for t in ts]

    cases = {
        "fm":      (lambda a: F_.ctr_logit(a[0], a[1], bias=a[4].reshape(1, 1)),
                    lambda a: a[0].sum(1, keepdim=True) + a[1].sum(1) + a[4].view(1, 1), (0, 1, 4)),
        "deepfm":  (lambda a: F_.ctr_logit(a[0], a[1], [a[2]]),
                    lambda a: torch.cat([a[0], a[1].reshape(B, -1)], 1).sum(1, keepdim=True) + a[2], (0, 1, 2)),
        "xdeepfm": (lambda a: F_.ctr_logit(None, a[1], [a[3], a[2]], bias=a[4]),
                    lambda a: a[1].sum(1) + a[3] + a[2] + a[4], (1, 2, 3, 4)),
    }
    for name, (ours, ref, used) in cases.items():
        a = leaves([fm, feat, deep, cin, bias], dev)
        r = leaves([t.float() for t in (fm, feat, deep, cin, bias)], "cpu")
        y = ours(a)
        yr = ref(r)
        assert y.shape == (B, 1) and y.dtype == dtype
        scale = float(sum(t.float().abs().sum(dim=tuple(range(1, t.dim()))).max() for t in (fm, feat)) + 3)
        assert float((y.detach().float().cpu() - yr.detach()).abs().max()) <= tol * scale, name       # a sum with cancellation: absolute scale
        (y.float() * go.to(dev)).sum().backward()
        (yr * go).sum().backward()
        for k in used:
            assert a[k].grad is not None, (name, k)
            got = a[k].grad.float().cpu().reshape(r[k].grad.shape)
            if k == 4:      # the bias gradient is a sum over the batch of rounded terms: judged on the scale of its terms
                assert float((got - r[k].grad).abs().max()) <= tol * float(go.abs().sum()), (name, "bias")
            else:
                assert rel_err(got, r[k].grad) <= tol, (name, k)


def test_ctr_logit_takes_a_strided_column_and_named_tensors(dev):
    """the logit column of the fused MLP tail is column 0 of an 8-wide padded output: read in place through its stride"""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(5)
    B = 300
    wide = torch.randn(B, 8, generator=g).to(dev)
    fm = torch.randn(B, 64, generator=g).to(dev).refine_names('B', 'O')
    feat = torch.randn(B, 39, 1, generator=g).to(dev).refine_names('B', 'N', 'E')
    y = F_.ctr_logit(fm, feat, [wide[:, :1]])
    ref = fm.rename(None).sum(1, keepdim=True) + feat.rename(None).sum(1) + wide[:, :1]
    assert rel_err(y.cpu(), ref.cpu()) <= 1e-5
    with pytest.raises(ValueError):
        F_.ctr_logit(fm, feat, [wide])
    with pytest.raises(ValueError):
        F_.ctr_logit()
    with pytest.raises(RuntimeError, match="no CPU path"):
        F_.ctr_logit(fm.cpu(), feat.cpu())


@pytest.mark.parametrize("ldtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B", [1, 255, 8192, 8193, 65536, 300001])   # 8192: the last one-workgroup batch
def test_bce_with_logits_vs_aten(dev, dtype, tol, ldtype, B):
    from torecsys_amd import functional as F_
    from torecsys_amd.fused import BCEWithLogitsLoss
    g = torch.Generator().manual_seed(B)
    x = (4 * torch.randn(B, 1, generator=g)).to(dtype)
    x[0] = 60.0                                    # saturated logits: the stable form must not overflow
    if B > 1:
        x[1] = -60.0
    y = (torch.rand(B, 1, generator=g) < 0.25).to(ldtype)
    xd = x.to(dev).requires_grad_()
    loss = BCEWithLogitsLoss()(xd, y.to(dev))
    assert loss.dtype == torch.float32 and loss.dim() == 0
    (loss * 3.0).backward()
    xr = x.float().requires_grad_()
    lr = O.bce_with_logits(xr, y)
    (lr * 3.0).backward()
    assert abs(float(loss) - float(lr)) <= 1e-5 * max(1.0, abs(float(lr)))     # fp32 accumulation in both
    assert torch.isfinite(xd.grad).all()
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= tol
    # reproducible: partial sums are folded in a fixed order
    assert float(F_.bce_with_logits(xd.detach(), y.to(dev))) == float(loss)


def test_bce_module_refuses_what_it_does_not_cover():
    from torecsys_amd.fused import BCEWithLogitsLoss
    for kw in ({"reduction": "sum"}, {"pos_weight": torch.ones(1)}, {"weight": torch.ones(1)}):
        with pytest.raises(NotImplementedError):
            BCEWithLogitsLoss(**kw)


def test_deepfm_step_with_fused_head_and_loss_matches_unfused(dev, monkeypatch):
    """harness DeepFM + the fused loss against the same model with the ATen head / nn.BCEWithLogitsLoss: loss and every
    gradient (tables included: the head's backward hands broadcast VIEWS to the lookups' backwards)"""
    from harness import ctr_models as M
    from torecsys_amd.fused import BCEWithLogitsLoss
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    B, N, E = 1024, 7, 32
    sizes = [50, 3, 1000, 17, 400, 9, 121]
    torch.manual_seed(3)
    emb = MultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=True)
    feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
    emb.set_schema(["c0"]); feat.set_schema(["c0"])
    inputs = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev)
    model = M.DeepFactorizationMachineModel(E, N, [64, 32], fm_dropout_p=0.0).to(dev)
    params = list(inputs.parameters()) + list(model.parameters())
    g = torch.Generator().manual_seed(0)
    ix = torch.stack([torch.randint(0, s, (B,), generator=g) for s in sizes], 1).to(dev)
    lab = (torch.rand(B, 1, generator=g) < 0.3).float().to(dev)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(M, "FUSED_HEAD", fused)
        crit = BCEWithLogitsLoss() if fused else torch.nn.BCEWithLogitsLoss()
        for p in params:
            p.grad = None
        loss = crit(model(**inputs({"c0": ix})), lab)
        loss.backward()
        res.append((float(loss), [p.grad.clone() for p in params]))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[0][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert rel_err(b.cpu(), a.cpu()) <= 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("B,N,Ea,Eb", [(257, 39, 64, 64), (1, 1, 8, 8), (1000, 10, 16, 64), (33, 5, 24, 8), (4096, 39, 64, 64),
                                       (70000, 3, 128, 32), (5, 16, 256, 256)])
def test_cat_head_vs_the_concatenated_linear(dev, dtype, tol, B, N, Ea, Eb):
    """F_.cat_head = nn.Linear(N*(Ea+Eb), 1) on cat((a, d), dim=2).flatten(1) (deep_and_cross_network.py:82-92): output
    and the gradients of both blocks, the weight row and the bias, against the composition in fp32 on the same (rounded)
    operands.  The block gradients are g[r] * w rounded once: equal to the composition's bit for bit in fp32."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B + N + Ea + Eb)
    C = N * (Ea + Eb)
    a = torch.randn(B, N, Ea, generator=g).to(dtype)
    d = torch.randn(B, N, Eb, generator=g).to(dtype)
    w = (torch.randn(1, C, generator=g) / C ** 0.5).to(dtype)
    b = torch.randn(1, generator=g).to(dtype)
    go = torch.randn(B, 1, generator=g).to(dtype)
    da, dd, dw, db = (t.to(dev).clone().requires_grad_() for t in (a, d, w, b))
    if C * a.element_size() // 16 > 1024:       # more 16-byte vectors per sample than a wave's lanes hold (fp32, 39 x 128)
        assert not F_.cat_head_supported(da, dd, dw)
        with pytest.raises(RuntimeError):
            F_.cat_head(da, dd, dw, db)
        return
    assert F_.cat_head_supported(da, dd, dw)
    out = F_.cat_head(da, dd, dw, db)
    out.backward(go.to(dev))
    ra, rd, rw, rb = (t.float().clone().requires_grad_() for t in (a, d, w, b))
    ref = torch.nn.functional.linear(torch.cat((ra, rd), dim=2).reshape(B, -1), rw, rb)
    ref.backward(go.float())
    assert out.shape == (B, 1) and out.dtype == dtype
    assert rel_err(out.float().cpu(), ref.detach()) <= tol
    assert rel_err(da.grad.float().cpu(), ra.grad) <= tol
    assert rel_err(dd.grad.float().cpu(), rd.grad) <= tol
    assert rel_err(dw.grad.float().cpu(), rw.grad) <= tol
    assert rel_err(db.grad.float().cpu(), rb.grad) <= tol
    if dtype == torch.float32:
        assert torch.equal(da.grad.cpu(), ra.grad) and torch.equal(dd.grad.cpu(), rd.grad)


def test_cat_head_partial_gradients_and_rejections(dev):
    """only the blocks' gradients (frozen head), only the head's (detached blocks); shapes the kernel does not take are
    reported by cat_head_supported and refused by the C entry point"""
    from torecsys_amd import _abi, functional as F_
    torch.manual_seed(3)
    a = torch.randn(64, 7, 16, device=dev)
    d = torch.randn(64, 7, 8, device=dev)
    w = torch.randn(1, 7 * 24, device=dev)
    b = torch.randn(1, device=dev)
    go = torch.randn(64, 1, device=dev)
    a1, d1 = a.clone().requires_grad_(), d.clone().requires_grad_()
    F_.cat_head(a1, d1, w, b).backward(go)
    wa = w.view(7, 24)[:, :16].reshape(1, 7, 16)
    assert torch.equal(a1.grad, go.view(64, 1, 1) * wa)
    w2, b2 = w.clone().requires_grad_(), b.clone().requires_grad_()
    F_.cat_head(a, d, w2, b2).backward(go)
    ref_w = (go.view(64, 1) * torch.cat((a, d), 2).reshape(64, -1)).sum(0, keepdim=True)
    assert rel_err(w2.grad.cpu(), ref_w.cpu()) <= 1e-5 and rel_err(b2.grad.cpu(), go.sum().view(1).cpu()) <= 1e-5
    assert not F_.cat_head_supported(a[:, :, :6], d, w)                      # 24-byte rows
    assert not F_.cat_head_supported(a, d, w[:, :-1])
    out = torch.empty(64, 1, device=dev)
    with pytest.raises(RuntimeError):
        _abi.call("trs_cat_head_fwd", _abi.ptr(a), _abi.ptr(d), _abi.ptr(w), _abi.ptr(b), 64, 7, 6, 8, 0, _abi.ptr(out),
                  _abi.stream_ptr())


def test_patched_heads_accept_an_fp32_bias_under_bf16_activations(dev):
    """patch(heads=True) wraps the reference's model forwards.  The reference's `outputs += self.bias` promotes in place,
    so bf16 tables / activations with the default fp32 bias parameter work there: the wrapped forwards must take that
    case too (bias cast to the activation dtype, gradient back in the parameter's own dtype) instead of raising from
    ctr_logit's same-dtype check.  Stand-in classes shaped like models/ctr/factorization_machine.py:42-71 and
    xdeep_fm.py:82-124 (the reference package is absent on the GPU box)."""
    import torch.nn as nn
    from torecsys_amd import layers as L
    from torecsys_amd import patching as P
    B, N, E = 512, 10, 16
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(B, N, E, generator=g).to(torch.bfloat16)
    feat = torch.randn(B, N, 1, generator=g).to(torch.bfloat16)

    class FactorizationMachineModel(nn.Module):
        def __init__(self):
            super().__init__()
            self.use_bias = True
            self.fm = L.FactorizationMachineLayer(0.0)
            self.bias = nn.Parameter(torch.full((1, 1), 0.37))          # fp32, as nn.Parameter(torch.zeros) leaves it

        def forward(self, feat_inputs, emb_inputs):
            raise AssertionError("the reference forward must not be needed for this case")

    FactorizationMachineModel.forward = P._fm_forward(FactorizationMachineModel.forward)
    m = FactorizationMachineModel().to(dev)
    e, f = emb.to(dev).requires_grad_(), feat.to(dev).requires_grad_()
    out = m(f, e)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (B, 1)
    ref = O.fm_model(feat.float(), emb.float(), torch.full((1, 1), 0.37))
    assert rel_err(out.float().cpu(), ref) <= 1e-2
    out.float().sum().backward()
    assert m.bias.grad is not None and m.bias.grad.dtype == torch.float32
    assert abs(float(m.bias.grad) - B) <= 1e-2 * B

    class XDeepFactorizationMachineModel(nn.Module):
        def __init__(self):
            super().__init__()
            self.cin = L.CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=1, layer_sizes=[8, 8])
            self.deep = L.MultilayerPerceptionLayer(inputs_size=N * E, output_size=1, layer_sizes=[32])
            self.bias = nn.Parameter(torch.full((1,), -0.25))

        def forward(self, feat_inputs, emb_inputs):
            raise AssertionError("the reference forward must not be needed for this case")

    XDeepFactorizationMachineModel.forward = P._xdeepfm_forward(XDeepFactorizationMachineModel.forward)
    torch.manual_seed(5)
    x = XDeepFactorizationMachineModel().to(dev)
    x.cin.to(torch.bfloat16)
    x.deep.to(torch.bfloat16)          # the bias stays fp32
    x.eval()
    with torch.no_grad():
        got = x(feat.to(dev), emb.to(dev))
        parts = (P._plain(x.cin(emb.to(dev))).float() + P._plain(x.deep(emb.to(dev).reshape(B, -1))).float()
                 + feat.to(dev).float().sum(1) - 0.25)
    assert got.dtype == torch.bfloat16
    assert rel_err(got.float().cpu(), parts.cpu()) <= 1e-2
