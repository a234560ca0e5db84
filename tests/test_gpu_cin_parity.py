"""F4 / CIN on the bf16 matrix-core path, pinned to the oracle piece by piece (SURVEY.md section 8a F4,
compress_interaction_network.py:125-181) at north_star's bf16 tolerance (1e-2):

  * each contraction kernel ALONE -- trs_cin_cl_fwd, trs_cin_cl_bwd_data, trs_cin_dw -- against the oracle's contraction
    (``oracle.cin_contraction`` = outer product + Conv1d(k=1)) on the same bf16-rounded operands at xDeepFM's shapes
    (N 39, H in {39, 128}, C 256, E 64, B 4096).  One contraction has no ReLU mask, so nothing but rounding separates
    the two.  The CPU oracle gets a random sample of the batch rows (rows are independent); the whole batch is checked
    against the same contraction restated in float64 torch ops on the device;
  * the glue kernels ALONE (trs_cin_glue_*: BatchNorm1d in TRAIN mode + ReLU + chunk + sum over E) against
    ``oracle.cin_glue`` in fp32;
  * the whole layer end to end in train mode: forward against the oracle, and gradients against the oracle's gradients
    EVALUATED UNDER THE KERNEL'S OWN ReLU MASKS.  A ReLU mask is a discontinuity of the gradient: an activation that
    bf16 rounding moves across zero changes the gradient by a whole term, which no tolerance on continuous arithmetic
    covers.  The masks the kernel used are recomputed from its own contraction outputs and handed to the oracle as its
    activation (z = y * mask), which removes exactly that effect and nothing else.
"""
import pytest
import torch

from conftest import batch_sum_err, rel_err, rel_err_rows, sum_err
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

TOL = 1e-2            # north_star: 1e-2 relative for bf16 interaction sums


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _operands(B, N, H, C, E, seed):
    """bf16-rounded operands of one contraction; channels-first (reference) orientation on the CPU"""
    g = torch.Generator().manual_seed(seed)
    x0 = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16()
    xk = x0 if H == N else (0.5 * torch.randn(B, H, E, generator=g)).abs_().bfloat16()     # hidden states are post-ReLU
    W = (torch.randn(C, N * H, generator=g) / (N * H) ** 0.5).bfloat16()
    bias = (0.1 * torch.randn(C, generator=g)).bfloat16()
    gy = torch.randn(B, C, E, generator=g).bfloat16()
    return x0, xk, W, bias, gy


def _channels_last(x, ld):
    """(B,H,E) -> zero-padded (B,E,ld): the layout the matrix-core kernels read"""
    B, H, E = x.shape
    out = x.new_zeros(B, E, ld)
    out[:, :, :H] = x.transpose(1, 2)
    return out


def _pad32(n):
    return (n + 31) // 32 * 32


def _device_f64(x0, xk, W, bias, gy, same, chunk=128):
    """The same contraction (outer product over the field dims, then the channel-mixing GEMM,
    compress_interaction_network.py:125-137) in float64 torch ops on the device, whole batch in chunks.
    Returns y (B,C,E), dx0 (B,N,E), dxk (B,H,E), dW (C,N*H), db (C)."""
    B, N, E = x0.shape
    H = xk.shape[1]
    Wd = W.double().requires_grad_()
    bd = bias.double().requires_grad_()
    ys, g0, gk = [], [], []
    for s in range(0, B, chunk):
        a = x0[s:s + chunk].double().requires_grad_()
        k = a if same else xk[s:s + chunk].double().requires_grad_()
        z = (a.unsqueeze(2) * k.unsqueeze(1)).reshape(a.shape[0], N * H, E)
        y = torch.einsum("ck,bke->bce", Wd, z) + bd.view(1, -1, 1)
        (y * gy[s:s + chunk].double()).sum().backward()
        ys.append(y.detach())
        g0.append(a.grad)
        gk.append(a.grad if same else k.grad)
    return torch.cat(ys), torch.cat(g0), torch.cat(gk), Wd.grad, bd.grad


@pytest.mark.parametrize("B,N,H,C,E", [(4096, 39, 39, 256, 64), (4096, 39, 128, 256, 64), (1001, 10, 10, 64, 32),
                                       (1001, 10, 32, 128, 32), (515, 6, 6, 64, 16), (515, 6, 64, 32, 16),
                                       (130, 40, 64, 32, 128), (262, 26, 256, 64, 32), (96, 64, 64, 64, 64),
                                       # first-layer form (xk is x0, N > 32: folded weights, diagonal-and-below blocks)
                                       (300, 48, 48, 128, 64), (200, 33, 33, 256, 32), (64, 100, 100, 128, 64),
                                       (50, 36, 36, 128, 32)],
                         ids=lambda v: str(v))
def test_cin_contraction_kernels_alone(dev, B, N, H, C, E):
    """BASELINE shapes plus the narrow ones (one or two 16-pixel tiles per wave, partial last step, short samples)."""
    from torecsys_amd import functional as F_
    S = 32
    x0, xk, W, bias, gy = _operands(B, N, H, C, E, seed=100 + H + E)
    same = H == N
    x0T = _channels_last(x0, _pad32(N)).to(dev).requires_grad_()
    xkT = x0T if same else _channels_last(xk, _pad32(H)).to(dev).requires_grad_()
    Wd, bd = W.to(dev).requires_grad_(), bias.to(dev).requires_grad_()
    gyT = gy.transpose(1, 2).contiguous().to(dev)

    # ---- forward kernel alone
    yT = F_.cin_contract_cl(x0T, xkT, Wd, bd, N, H)
    assert yT.shape == (B, E, C) and yT.dtype == torch.bfloat16
    y = yT.detach().transpose(1, 2).float()                                         # (B,C,E)

    # ---- data-gradient kernel alone (no weight gradient requested), then the weight-gradient kernel alone
    ins = (x0T,) if same else (x0T, xkT)
    gd = torch.autograd.grad(yT, ins, gyT, retain_graph=True)
    dx0 = gd[0].detach()[:, :, :N].transpose(1, 2).float()                          # (B,N,E); x0 == xk: the sum of both
    dxk = None if same else gd[1].detach()[:, :, :H].transpose(1, 2).float()
    if gd[0].shape[2] > N:
        assert float(gd[0][:, :, N:].float().abs().max()) == 0.0                    # padding columns get no gradient
    dW, db = torch.autograd.grad(yT, (Wd, bd), gyT, retain_graph=True)
    torch.cuda.synchronize()

    # ---- whole batch against float64 on the device
    ry, r0, rk, rW, rb = _device_f64(x0.to(dev), xk.to(dev), W.to(dev), bias.to(dev), gy.to(dev), same)
    assert rel_err(y, ry) <= TOL and rel_err_rows(y, ry) <= TOL
    assert rel_err(dx0, r0) <= TOL and rel_err_rows(dx0, r0) <= TOL
    if not same:
        assert rel_err(dxk, rk) <= TOL and rel_err_rows(dxk, rk) <= TOL
    assert rel_err(dW.float(), rW) <= TOL and rel_err_rows(dW.float(), rW) <= TOL
    assert rel_err(db.float(), rb) <= TOL

    # ---- sampled rows against the CPU oracle (rows of one contraction are independent)
    g = torch.Generator().manual_seed(7)
    rows = torch.randperm(B, generator=g)[:S].sort().values
    a = x0[rows].float().permute(0, 2, 1).requires_grad_()                          # (S,E,N), the oracle's orientation
    k = a if same else xk[rows].float().permute(0, 2, 1).requires_grad_()
    Wo = W.float().reshape(C, N * H, 1).requires_grad_()
    yo = O.cin_contraction(a, k, Wo, bias.float())
    (yo * gy[rows].float()).sum().backward()
    assert rel_err(y[rows.to(dev)].cpu(), yo.detach()) <= TOL
    assert rel_err_rows(y[rows.to(dev)].cpu(), yo.detach()) <= TOL
    assert rel_err_rows(dx0[rows.to(dev)].cpu(), a.grad.permute(0, 2, 1)) <= TOL
    if not same:
        assert rel_err_rows(dxk[rows.to(dev)].cpu(), k.grad.permute(0, 2, 1)) <= TOL
    # weight gradient by linearity: with the output gradient zeroed outside the sampled rows, dW is the oracle's dW
    gsel = torch.zeros_like(gyT)
    gsel[rows.to(dev)] = gyT[rows.to(dev)]
    (dWs,) = torch.autograd.grad(yT, (Wd,), gsel)
    assert rel_err(dWs.float().cpu(), Wo.grad.reshape(C, N * H)) <= TOL
    assert rel_err_rows(dWs.float().cpu(), Wo.grad.reshape(C, N * H)) <= TOL


@pytest.mark.parametrize("B,E,C,direct", [(512, 64, 256, False), (96, 64, 256, True), (300, 16, 64, False)])
def test_cin_glue_alone_train_mode_vs_oracle(dev, B, E, C, direct):
    """trs_cin_glue_* in TRAIN mode (batch statistics) against ``oracle.cin_glue`` in fp32 on the same bf16 input."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B + C)
    y0 = (torch.randn(B, E, C, generator=g) * 1.3 + 0.2).bfloat16()                 # channels-last contraction result
    gamma = (torch.rand(C, generator=g) + 0.5).bfloat16()
    beta = (0.2 * torch.randn(C, generator=g)).bfloat16()
    D, Hs = (C, 0) if direct else (C // 2, C // 2)
    bn = torch.nn.BatchNorm1d(C).to(dev).bfloat16().train()
    bn.weight.data.copy_(gamma)
    bn.bias.data.copy_(beta)
    ya = y0.to(dev).requires_grad_()
    hidden, pooled = F_.cin_glue(ya, bn, D, Hs)
    gh = torch.randn(hidden.shape, generator=g).bfloat16()
    gp = torch.randn(pooled.shape, generator=g).bfloat16()
    ((hidden.float() * gh.to(dev).float()).sum() + (pooled.float() * gp.to(dev).float()).sum()).backward()
    # oracle, channels-first fp32
    yr = y0.float().transpose(1, 2).contiguous().requires_grad_()                   # (B,C,E)
    gr, br = gamma.float().requires_grad_(), beta.float().requires_grad_()
    rm, rv = torch.zeros(C), torch.ones(C)
    z, direct_r, hidden_r = O.cin_glue(yr, gr, br, rm, rv, True, direct, True)
    pooled_r = direct_r.sum(dim=-1)
    ((hidden_r * gh.float()).sum() + (pooled_r * gp.float()).sum()).backward()
    assert rel_err(hidden.float().cpu(), hidden_r.detach()) <= TOL
    assert rel_err_rows(hidden.float().cpu(), hidden_r.detach()) <= TOL
    # pooled = a sum over E of non-negative terms: no cancellation, plain relative error
    assert rel_err(pooled.float().cpu(), pooled_r.detach()) <= TOL
    assert rel_err_rows(pooled.float().cpu(), pooled_r.detach()) <= TOL
    assert rel_err(ya.grad.float().cpu(), yr.grad.transpose(1, 2)) <= TOL
    assert rel_err(bn.weight.grad.float().cpu(), gr.grad) <= TOL
    assert rel_err(bn.bias.grad.float().cpu(), br.grad) <= TOL
    assert rel_err(bn.running_mean.float().cpu(), rm) <= TOL
    assert rel_err(bn.running_var.float().cpu(), rv) <= TOL


class _GlueRecorder:
    """Wraps functional.cin_glue: records every layer's contraction output (the glue kernels' input)."""

    def __init__(self, F_):
        self.F_, self.orig, self.seen = F_, F_.cin_glue, []

    def __enter__(self):
        def rec(yT, bn, D, Hs):
            self.seen.append((yT.detach(), bn, D, Hs))
            return self.orig(yT, bn, D, Hs)
        self.F_.cin_glue = rec
        return self

    def __exit__(self, *exc):
        self.F_.cin_glue = self.orig
        return False


def _kernel_masks(seen):
    """ReLU masks the glue kernels applied, recomputed from their own inputs: z = y*scale + shift > 0 with the
    train-mode batch statistics of y (fp64 reduction), channels-first (B,C,E) on the CPU."""
    masks = []
    for yT, bn, D, Hs in seen:
        y = yT.double()
        mean = y.mean(dim=(0, 1))
        var = y.var(dim=(0, 1), unbiased=False)
        scale = bn.weight.double() / torch.sqrt(var + bn.eps)
        shift = bn.bias.double() - mean * scale
        z = yT.float() * scale.float() + shift.float()
        masks.append((z > 0).float().transpose(1, 2).contiguous().cpu())
    return masks


@pytest.mark.parametrize("B,N,E,sizes,direct", [(256, 39, 64, [128, 128, 128], False), (70, 39, 64, [64, 64], False),
                                                (33, 10, 32, [32, 64, 32], False), (16, 39, 64, [128, 128], False),
                                                (20, 6, 16, [32, 64], True), (9, 40, 128, [32], False)])
def test_cin_layer_mfma_vs_oracle_under_kernel_masks(dev, B, N, E, sizes, direct):
    from torecsys_amd import functional as F_
    from torecsys_amd.layers import CompressInteractionNetworkLayer
    torch.manual_seed(B + N + E)
    O_SIZE = 2
    lay = CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=O_SIZE, layer_sizes=list(sizes),
                                          is_direct=direct)
    for seq in lay.model:                       # non-trivial affine BatchNorm parameters
        seq.Batchnorm.weight.data.uniform_(0.5, 1.5)
        seq.Batchnorm.bias.data.normal_(0.0, 0.2)
    lay = lay.to(dev).bfloat16().train()
    g = torch.Generator().manual_seed(5)
    x = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16()
    xd = x.to(dev).requires_grad_()
    with _GlueRecorder(F_) as rec:
        y = lay(xd)
    assert len(rec.seen) == len(sizes), "the layer did not take the channels-last matrix-core path"
    go = torch.randn(B, O_SIZE, generator=g)
    (y.rename(None).float() * go.to(dev)).sum().backward()
    masks = _kernel_masks(rec.seen)

    f32 = lambda t: t.detach().float().cpu()
    P = dict(conv_weights=[f32(s.Conv1d.weight).requires_grad_() for s in lay.model],
             conv_biases=[f32(s.Conv1d.bias).requires_grad_() for s in lay.model],
             bn_weights=[f32(s.Batchnorm.weight).requires_grad_() for s in lay.model],
             bn_biases=[f32(s.Batchnorm.bias).requires_grad_() for s in lay.model],
             fc_weight=f32(lay.fc.weight).requires_grad_(), fc_bias=f32(lay.fc.bias).requires_grad_())
    # forward: the oracle as it is (ReLU)
    xr = x.float().requires_grad_()
    yr, inter, pooled = O.cin_layer(xr, **P, is_direct=direct, training=True, return_intermediates=True)
    terms = pooled.detach().abs() @ P["fc_weight"].detach().abs().t() + P["fc_bias"].detach().abs()
    # fc(pooled) sums 100s of signed terms: the error is bounded relative to the magnitude of what is summed
    assert sum_err(f32(y.rename(None)), yr.detach(), terms) <= TOL
    # how many activations sit on the other side of zero in the two pipelines (reported, bounded loosely)
    flips = sum(float(((z > 0).float() != m).float().mean()) for (_, z), m in zip(inter, masks)) / len(masks)
    assert flips <= 2e-2, flips
    # backward: the oracle's gradients under the kernel's masks
    acts = [(lambda t, m=m: t * m) for m in masks]
    xm = x.float().requires_grad_()
    ym, inter_m, _ = O.cin_layer(xm, **P, is_direct=direct, training=True, activation=acts, return_intermediates=True)
    for _, z in inter_m:
        z.retain_grad()
    (ym * go).sum().backward()
    assert rel_err(f32(xd.grad), xm.grad) <= TOL
    assert rel_err_rows(f32(xd.grad), xm.grad, floor_frac=5e-2) <= 2 * TOL
    for k, seq in enumerate(lay.model):
        assert rel_err(f32(seq.Conv1d.weight.grad), P["conv_weights"][k].grad) <= TOL, k
        # BatchNorm gamma / beta gradients: sums over (B, E) of signed terms dz * zhat and dz (see batch_sum_err)
        pre, z = inter_m[k]
        dz = z.grad * masks[k]
        zhat = (pre - pre.mean(dim=(0, 2), keepdim=True)) / torch.sqrt(pre.var(dim=(0, 2), unbiased=False, keepdim=True) + 1e-5)
        assert batch_sum_err(f32(seq.Batchnorm.weight.grad), P["bn_weights"][k].grad,
                             ((dz * zhat.detach()) ** 2).sum(dim=(0, 2)), TOL) <= 1.0, k
        assert batch_sum_err(f32(seq.Batchnorm.bias.grad), P["bn_biases"][k].grad, (dz ** 2).sum(dim=(0, 2)), TOL) <= 1.0, k
    assert rel_err(f32(lay.fc.weight.grad), P["fc_weight"].grad) <= TOL
    # fc.bias.grad = sum_b of the bf16-rounded output gradient (nn.Linear's own backward, no kernel of this path)
    assert batch_sum_err(f32(lay.fc.bias.grad), P["fc_bias"].grad, (go ** 2).sum(dim=0), TOL) <= 1.0


@pytest.mark.parametrize("B,N,E,ld", [(65, 39, 64, 64), (1, 1, 8, 8), (300, 10, 16, 32), (4100, 33, 64, 64), (7, 64, 64, 64),
                                      (129, 5, 40, 8)])
def test_transpose_pad_is_the_padded_transposition_bit_for_bit(dev, B, N, E, ld):
    """F_.transpose_pad = new_zeros(B,E,ld)[:, :, :N] = x.transpose(1,2) (the CIN layer's channels-last entry,
    compress_interaction_network.py:105) and its gradient = the un-padded transposition back: moves only, so equal bit
    for bit; more samples than workgroups on the (4100, ...) case"""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(B * N + E)
    x = torch.randn(B, N, E, generator=g).to(torch.bfloat16).to(dev).requires_grad_()
    assert F_.transpose_pad_supported(x, ld)
    y = F_.transpose_pad(x, ld)
    ref = x.detach().new_zeros(B, E, ld)
    ref[:, :, :N] = x.detach().transpose(1, 2)
    assert y.shape == (B, E, ld) and torch.equal(y, ref)
    go = torch.randn(B, E, ld, generator=g).to(torch.bfloat16).to(dev)
    y.backward(go)
    assert torch.equal(x.grad, go[:, :, :N].transpose(1, 2).contiguous())
    assert not F_.transpose_pad_supported(x, 128) and not F_.transpose_pad_supported(x.float(), ld)


@pytest.mark.parametrize("B,N,E,sizes", [(256, 39, 64, [128, 128, 128]), (64, 39, 64, [64, 128]), (96, 10, 32, [32, 64])])
@pytest.mark.parametrize("train", [True, False])
def test_last_layer_backward_skips_the_dead_half_exactly(dev, monkeypatch, B, N, E, sizes, train):
    """The last CIN layer's "hidden" half is computed, split off and never used (compress_interaction_network.py:151-156,
    176-181), so its gradient is exactly zero; the backward of that layer's contraction leaves those channels out
    (trs_cin_cl_bwd_data_live / trs_cin_dw_live).  Leaving out exact zeros must change NOTHING: every gradient of the layer
    -- input, all convolution weights and biases, BatchNorm affine parameters, fc -- bit-identical with the switch off,
    and the dead rows of the last convolution's weight gradient exactly zero."""
    from torecsys_amd import functional as F_
    from torecsys_amd.layers import CompressInteractionNetworkLayer
    torch.manual_seed(B + N)
    lay = CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=1, layer_sizes=sizes).to(dev).bfloat16()
    lay.train(train)
    x0 = (0.5 * torch.randn(B, N, E)).bfloat16().to(dev)
    gy = torch.randn(B, 1).bfloat16().to(dev)
    res = []
    for skip in (True, False):
        monkeypatch.setattr(F_, "CIN_SKIP_DEAD", skip)
        for p in lay.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        state = {k: v.clone() for k, v in lay.state_dict().items()}
        y = lay(x)
        y.rename(None).backward(gy)
        torch.cuda.synchronize()
        res.append(([x.grad.clone()] + [p.grad.clone() for p in lay.parameters()], y.rename(None).detach().clone()))
        lay.load_state_dict(state)            # (train mode moved the running statistics: same start for the second run)
    assert torch.equal(res[0][1], res[1][1])
    names = ["input"] + [n for n, _ in lay.named_parameters()]
    last_w = f"model.{len(sizes) - 1}.Conv1d.weight"
    for n, a, b in zip(names, res[0][0], res[1][0]):
        if n == last_w:
            # the one tensor whose SUMMATION ORDER changes: half as many channel blocks -> twice as many sample ranges in
            # the split-K weight gradient (fp32 partials added in another order, then one bf16 rounding)
            assert rel_err(a.float().cpu(), b.float().cpu()) <= 4e-3, n
        else:
            assert torch.equal(a, b), n
    last = lay.model[-1].Conv1d
    C = last.out_channels
    assert float(last.weight.grad[C // 2:].float().abs().max()) == 0.0
    assert float(last.weight.grad[: C // 2].float().abs().max()) > 0.0
