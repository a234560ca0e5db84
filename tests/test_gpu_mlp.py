"""The MLP stack (layers/ctr/multilayer_perceptron.py:24-84) stays on hipBLASLt GEMMs, but on the HIP device with
bf16 parameters they are arranged differently (zero-padded hidden widths, bias+ReLU epilogue, split-K weight
gradient, fused ReLU-backward + bias gradient).  Results must match a plain torch.nn stack."""
import copy

import pytest
import torch
import torch.nn as nn

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _plain(mlp):
    return copy.deepcopy(mlp.model)


@pytest.mark.parametrize("rows,shape3d", [(8192, True), (4096, False), (512, True)])
@pytest.mark.parametrize("sizes", [[400, 400, 400], [200, 136], [512, 300]])
def test_mlp_bf16_matches_plain_stack(dev, rows, shape3d, sizes):
    from torecsys_amd.layers import MultilayerPerceptionLayer, _pad_width
    torch.manual_seed(1)
    K = 96
    mlp = MultilayerPerceptionLayer(K, 3, sizes).to(dev).bfloat16()
    ref = _plain(mlp)
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    if shape3d:
        x = x.view(rows // 4, 4, K)
    xa = x.clone().requires_grad_()
    xb = x.clone().requires_grad_()
    ya = mlp(xa).rename(None)
    yb = ref(xb)
    assert ya.shape == yb.shape
    assert rel_err(ya.float().cpu(), yb.float().cpu()) <= 1e-2
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    assert rel_err(xa.grad.float().cpu(), xb.grad.float().cpu()) <= 1e-2
    for (n, pa), (_, pb) in zip(mlp.model.named_parameters(), ref.named_parameters()):
        assert pa.grad.shape == pb.grad.shape == pa.shape, n
        assert pa.grad.is_contiguous()
        assert rel_err(pa.grad.float().cpu(), pb.grad.float().cpu()) <= 1e-2, n
    padded = any('_trs_padded' in m.__dict__ for m in mlp.model if isinstance(m, nn.Linear))
    expect = rows >= 4096 and any(_pad_width(s) != s for s in sizes)
    assert padded == expect


def test_padded_weights_follow_parameter_updates(dev):
    from torecsys_amd.layers import MultilayerPerceptionLayer
    torch.manual_seed(2)
    mlp = MultilayerPerceptionLayer(64, 1, [400, 400]).to(dev).bfloat16()
    x = torch.randn(4096, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
    y0 = mlp(x).rename(None).detach().clone()
    with torch.no_grad():
        for p in mlp.parameters():
            p.mul_(0.5)                       # in-place update, like an optimizer step
    y1 = mlp(x).rename(None).detach()
    ref = _plain(mlp)
    assert rel_err(y1.float().cpu(), ref(x).detach().float().cpu()) <= 1e-2
    assert not torch.equal(y0, y1)
    sd = {k: torch.randn_like(v) * 0.05 for k, v in mlp.state_dict().items()}
    mlp.load_state_dict(sd)                   # copy_ into the parameters: version bump
    assert rel_err(mlp(x).rename(None).detach().float().cpu(), _plain(mlp)(x).detach().float().cpu()) <= 1e-2
    assert set(mlp.state_dict().keys()) == set(sd.keys())
    assert mlp.model.Linear_0.weight.shape == (400, 64)
    # writes through .data do NOT bump _version (p.data.add_-style optimizers, clipping, EMA swap-in, broadcasts):
    # the padded copies must follow them as well
    for p in mlp.parameters():
        v = p._version
        p.data.mul_(-1.5)
        assert p._version == v
    assert rel_err(mlp(x).rename(None).detach().float().cpu(), _plain(mlp)(x).detach().float().cpu()) <= 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(65536, 512), (8192, 400), (1000, 64), (7, 8), (1, 128), (4099, 256)])
def test_rowdot_matches_linear(dev, dtype, rows, C):
    """one-output Linear as a row-wise dot product (trs_rowdot_fwd / trs_rowdot_bwd) against fp32 torch math"""
    from torecsys_amd import functional as F_
    torch.manual_seed(rows + C)
    h = torch.randn(rows, C, device=dev, dtype=dtype)
    w = (torch.randn(1, C, device=dev) / C ** 0.5).to(dtype)
    b = torch.randn(1, device=dev).to(dtype)
    if not F_.rowdot_supported(h):
        pytest.skip("row length not a power-of-two number of 16-byte vectors")
    ha, wa, ba = h.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    out = F_._RowDot.apply(ha, wa, ba, None, None)
    hf, wf, bf = h.float().requires_grad_(), w.float().requires_grad_(), b.float().requires_grad_()
    ref = hf @ wf.t() + bf
    assert out.shape == (rows, 1) and out.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(out.float().cpu(), ref.detach().cpu()) <= tol
    g = torch.randn(rows, 1, device=dev)
    out.backward(g.to(dtype))
    ref.backward(g.to(dtype).float())
    assert rel_err(ha.grad.float().cpu(), hf.grad.cpu()) <= tol
    assert wa.grad.shape == w.shape and ba.grad.shape == b.shape
    assert rel_err(wa.grad.float().cpu(), wf.grad.cpu()) <= tol
    assert abs(float(ba.grad.float()) - float(bf.grad)) <= tol * max(1.0, abs(float(bf.grad)), rows ** 0.5)


def test_mlp_logit_layer_uses_rowdot(dev):
    """DeepFM's deep tower ends in Linear(400, 1): that layer must run as trs_rowdot (and match the plain stack)"""
    from torecsys_amd import _abi
    from torecsys_amd.layers import MultilayerPerceptionLayer
    torch.manual_seed(5)
    mlp = MultilayerPerceptionLayer(128, 1, [400, 400]).to(dev).bfloat16()
    ref = _plain(mlp)
    x = torch.randn(8192, 128, device=dev, dtype=torch.bfloat16)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    _abi.time_kernel("trs_rowdot_fwd", True)
    try:
        ya = mlp(xa).rename(None)
        assert len(_abi.kernel_times_ms("trs_rowdot_fwd")) == 1
    finally:
        _abi.time_kernel("trs_rowdot_fwd", False)
    yb = ref(xb)
    assert rel_err(ya.float().cpu(), yb.float().cpu()) <= 1e-2
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    assert rel_err(xa.grad.float().cpu(), xb.grad.float().cpu()) <= 1e-2
    for (n, pa), (_, pb) in zip(mlp.model.named_parameters(), ref.named_parameters()):
        assert rel_err(pa.grad.float().cpu(), pb.grad.float().cpu()) <= 1e-2, n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("S,R,Cc,out_rows,out_cols,with_bias", [
    (32, 512, 2496, 400, 2496, True), (8, 512, 512, 400, 400, True), (4, 64, 96, 64, 96, False),
    (5, 128, 40, 100, 33, True), (1, 16, 300, 16, 300, True),
    # few rows, many splits (the 400 -> 1 head at 65 536 rows): the kernel with 16 threads per column
    (256, 8, 400, 1, 400, True), (70, 8, 33, 3, 33, True), (64, 8, 1000, 8, 999, False)])
def test_wgrad_finish_matches_torch(dev, dtype, S, R, Cc, out_rows, out_cols, with_bias):
    """trs_wgrad_finish = part.sum(0)[:out_rows, :out_cols].to(dtype) (+ the bias-gradient cast) in one launch"""
    from torecsys_amd import _abi
    torch.manual_seed(S * R + Cc)
    part = torch.randn(S, R, Cc, device=dev)
    gbf = torch.randn(R, device=dev)
    gw = torch.empty(out_rows, out_cols, dtype=dtype, device=dev)
    gb = torch.empty(out_rows, dtype=dtype, device=dev) if with_bias else None
    _abi.call("trs_wgrad_finish", _abi.ptr(part), S, R, Cc, out_rows, out_cols, _abi.value_dtype_code(gw), _abi.ptr(gw),
              _abi.ptr(gbf) if with_bias else _abi.ptr(None), _abi.ptr(gb), _abi.stream_ptr())
    ref = part.sum(0)[:out_rows, :out_cols]
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert rel_err(gw.float().cpu(), ref.cpu()) <= tol
    if with_bias:
        assert torch.equal(gb.cpu(), gbf[:out_rows].to(dtype).cpu())


def test_wgrad_finish_rejects_bad_arguments(dev):
    from torecsys_amd import _abi
    part = torch.zeros(2, 8, 8, device=dev)
    gw = torch.empty(8, 8, device=dev)
    with pytest.raises(RuntimeError):          # out_rows > R
        _abi.call("trs_wgrad_finish", _abi.ptr(part), 2, 8, 8, 9, 8, 0, _abi.ptr(gw), _abi.ptr(None), _abi.ptr(None),
                  _abi.stream_ptr())
    with pytest.raises(RuntimeError):          # gb without gb_f32
        _abi.call("trs_wgrad_finish", _abi.ptr(part), 2, 8, 8, 8, 8, 0, _abi.ptr(gw), _abi.ptr(None), _abi.ptr(gw),
                  _abi.stream_ptr())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("S,Cc,R,out_rows,out_cols,with_bias", [
    (16, 2496, 512, 400, 2496, True), (4, 96, 64, 64, 96, False), (5, 40, 128, 100, 33, True), (1, 300, 16, 16, 300, True)])
def test_wgrad_finish_transposed_matches_torch(dev, dtype, S, Cc, R, out_rows, out_cols, with_bias):
    """trs_wgrad_finish_t: part (S, Cc, R) holds slices of x^T g; gw = part.sum(0).t()[:out_rows, :out_cols]"""
    from torecsys_amd import _abi
    torch.manual_seed(S * R + Cc)
    part = torch.randn(S, Cc, R, device=dev)
    gbf = torch.randn(R, device=dev)
    gw = torch.empty(out_rows, out_cols, dtype=dtype, device=dev)
    gb = torch.empty(out_rows, dtype=dtype, device=dev) if with_bias else None
    _abi.call("trs_wgrad_finish_t", _abi.ptr(part), S, Cc, R, out_rows, out_cols, _abi.value_dtype_code(gw),
              _abi.ptr(gw), _abi.ptr(gbf) if with_bias else _abi.ptr(None), _abi.ptr(gb), _abi.stream_ptr())
    ref = part.sum(0).t()[:out_rows, :out_cols]
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert rel_err(gw.float().cpu(), ref.cpu()) <= tol
    if with_bias:
        assert torch.equal(gb.cpu(), gbf[:out_rows].to(dtype).cpu())


def test_mlp_wide_first_layer_uses_transposed_split(dev):
    """a 2496-wide first layer at 16 384 rows takes the x^T g split (trs_wgrad_finish_t) and still matches nn.Linear"""
    from torecsys_amd import _abi
    from torecsys_amd.layers import MultilayerPerceptionLayer
    torch.manual_seed(11)
    mlp = MultilayerPerceptionLayer(2496, 1, [400, 400]).to(dev).bfloat16()
    ref = _plain(mlp)
    x = (torch.randn(16384, 2496, device=dev) * 0.5).bfloat16()
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = mlp(xa).rename(None), ref(xb)
    g = torch.randn_like(yb)
    _abi.time_kernel("trs_wgrad_finish_t", True)
    try:
        ya.backward(g)
        assert len(_abi.kernel_times_ms("trs_wgrad_finish_t")) == 1
    finally:
        _abi.time_kernel("trs_wgrad_finish_t", False)
    yb.backward(g)
    assert rel_err(xa.grad.float().cpu(), xb.grad.float().cpu()) <= 1e-2
    for (n, pa), (_, pb) in zip(mlp.model.named_parameters(), ref.named_parameters()):
        assert pa.grad.shape == pb.grad.shape
        assert rel_err(pa.grad.float().cpu(), pb.grad.float().cpu()) <= 1e-2, n


@pytest.mark.parametrize("out_f", [1, 3, 10])
@pytest.mark.parametrize("rows,K,sizes", [(65536, 2496, [400, 400, 400]), (8192, 96, [400, 400]), (8192, 64, [400, 400, 400])])
def test_mlp_public_output_is_contiguous(dev, out_f, rows, K, sizes):
    """A user-facing MLP layer returns a contiguous (rows, out_f) tensor like the reference's DNNLayer (`.view` works for
    every out_f, no padded buffer behind it); only inside layers.strided_outputs() -- the fused head's own call -- may it
    hand back the strided view of its padded output."""
    from torecsys_amd.layers import MultilayerPerceptionLayer, strided_outputs
    torch.manual_seed(2)
    mlp = MultilayerPerceptionLayer(K, out_f, sizes).to(dev).bfloat16()
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16, requires_grad=True)
    y = mlp(x)
    assert y.names == ('B', 'O') and tuple(y.shape) == (rows, out_f)
    p = y.rename(None)
    assert p.is_contiguous()
    assert p.view(-1).shape[0] == rows * out_f
    with strided_outputs():
        ys = mlp(x).rename(None)
    assert torch.equal(ys.float().cpu(), p.float().cpu())
    p.float().sum().backward()
    assert x.grad is not None and mlp.model[0].weight.grad is not None
