"""the per-field MLP (DeepAndCrossNetwork's deep branch, models/ctr/deep_and_cross_network.py:71-87 ->
layers/ctr/multilayer_perceptron.py:63-84) as one HIP kernel per direction (trs_mlp_fused_*), against the oracle's
``mlp`` in fp32 on the same bf16-rounded parameters: output, input gradient, every weight / bias gradient."""
import pytest
import torch

from conftest import rel_err, rel_err_rows
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

TOL = 1e-2            # north_star: 1e-2 relative for bf16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 2, 3], ids=["tile-in-LDS kernels", "row-owner kernels", "row-owner forward + tile backward"])
def ro_mode(request):
    """The two stack shapes of the models have a second pair of kernels (csrc/mlp_ro.hpp), chosen by AUTO from 131 072
    rows on.  Which family runs is a per-call request (functional.mlp_family; trs_mlp_fused_family in the ABI): every test
    that takes this fixture issues its forwards under TILE and under ROW_OWNER (stacks the row-owner kernels do not cover
    fall back to the tile kernels)."""
    from torecsys_amd import functional as F_
    with F_.mlp_family(request.param):
        yield request.param


def _params(widths, g):
    Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).bfloat16() for i, o in zip(widths[:-1], widths[1:])]
    bs = [(0.1 * torch.randn(o, generator=g)).bfloat16() for o in widths[1:]]
    return Ws, bs


@pytest.mark.parametrize("shape,widths", [((700, 7), [64, 400, 400, 400, 64]),      # the DCN stack, ragged last tile
                                          ((70019,), [64, 400, 400, 400, 64]),     # ... more passes than workgroups
                                          ((257,), [64, 400, 400, 400, 64]),       # ... one row into the second pass
                                          ((1,), [64, 400, 400, 400, 64]),
                                          ((66003,), [416, 400, 400, 8]),          # the tail of a 400-400-400 deep branch
                                          ((4096,), [16, 72, 8]),                  # widths that are not multiples of 32
                                          ((33, 130), [64, 512, 64]),              # the widest supported layer
                                          ((5000,), [32, 104, 200, 40]),
                                          ((4224,), [128, 96, 96, 96, 96, 96, 24])])
def test_fused_mlp_vs_oracle(dev, shape, widths, ro_mode, monkeypatch):
    from torecsys_amd import functional as F_
    monkeypatch.setattr(F_, "FUSED_MLP_MIN_ROWS", 1)      # (the layers take the GEMM path below 4096 rows; the kernels do not care)
    g = torch.Generator().manual_seed(sum(widths) + shape[0])
    Ws, bs = _params(widths, g)
    x = torch.randn(*shape, widths[0], generator=g).bfloat16()
    gy = torch.randn(*shape, widths[-1], generator=g).bfloat16()
    xd = x.to(dev).requires_grad_()
    Wd = [w.to(dev).requires_grad_() for w in Ws]
    bd = [b.to(dev).requires_grad_() for b in bs]
    assert F_.mlp_fused_supported(xd, widths)
    y = F_.fused_mlp(xd, Wd, bd)
    assert y.shape == (*shape, widths[-1]) and y.dtype == torch.bfloat16
    y.backward(gy.to(dev))
    rows = lambda t: t.reshape(-1, t.shape[-1])
    # forward against the oracle as it is
    yo = O.mlp(x.float(), [w.float() for w in Ws], [b.float() for b in bs])
    assert rel_err(y.float().cpu(), yo) <= TOL
    # (the worst single row: over 66 000 rows of an 8-wide output the tail of the bf16 rounding noise reaches 2.3e-2)
    row_tol = (2 if y.numel() // y.shape[-1] < 50000 else 3) * TOL
    assert rel_err_rows(rows(y.float().cpu()), rows(yo), floor_frac=5e-2) <= row_tol
    # gradients against the oracle UNDER THE KERNEL'S OWN ReLU MASKS (a hidden unit whose pre-activation bf16
    # rounding moves across zero changes the gradient by a whole term -- see tests/test_gpu_cin_parity.py).  The sign
    # bits the forward kernel hands to the backward kernel are in the kernel's own order (opaque); they are the signs
    # of the hidden activations it stores, which is what the oracle is masked with here -- wrong bits would show up as
    # whole missing / extra terms in the gradients below
    _, hidden, masks, fam = F_.fused_mlp_forward_raw(rows(x).to(dev), [w.to(dev) for w in Ws], [b.to(dev) for b in bs])
    covered = widths in ([64, 400, 400, 400, 64], [416, 400, 400, 8])
    assert fam == (ro_mode if covered else F_.MLP_FAMILY_TILE)
    assert all(m.numel() == F_.size_query("trs_mlp_fused_mask_bytes", rows(x).shape[0]) for m in masks)
    unpacked = []
    for l in range(len(masks)):
        h = hidden[l].float().cpu()
        unpacked.append((h[:, :widths[l + 1]] > 0).float().reshape(*shape, widths[l + 1]))
        assert float(h[:, widths[l + 1]:].abs().max() if h.shape[1] > widths[l + 1] else 0.0) == 0.0
    it = iter(unpacked)
    xr = x.float().requires_grad_()
    Wr = [w.float().requires_grad_() for w in Ws]
    br = [b.float().requires_grad_() for b in bs]
    yr = O.mlp(xr, Wr, br, activation=lambda t: t * next(it))
    yr.backward(gy.float())
    flips = sum(float(((a > 0).float() != m).float().mean()) for a, m in zip(
        [torch.relu(torch.nn.functional.linear(x.float(), Ws[0].float(), bs[0].float()))], unpacked[:1]))
    assert flips <= 2e-2
    assert rel_err(xd.grad.float().cpu(), xr.grad) <= TOL
    assert rel_err_rows(rows(xd.grad.float().cpu()), rows(xr.grad), floor_frac=5e-2) <= row_tol
    for l, (a, b) in enumerate(zip(Wd, Wr)):
        assert rel_err(a.grad.float().cpu(), b.grad) <= TOL, ("dW", l)
    for l, (a, b) in enumerate(zip(bd, br)):
        assert rel_err(a.grad.float().cpu(), b.grad) <= TOL, ("db", l)


def test_mlp_layer_takes_fused_path_and_matches_gemm_path(dev):
    """MultilayerPerceptionLayer on a (B,N,E) block: the fused kernels and the hipBLASLt arrangement they replace give
    the same numbers (both bf16; compared with each other at 2e-2 and each with the oracle above / in test_gpu_mlp.py)."""
    from torecsys_amd import functional as F_
    from torecsys_amd.layers import DNNLayer
    torch.manual_seed(5)
    lay = DNNLayer(inputs_size=64, output_size=64, layer_sizes=[400, 400, 400]).to(dev).bfloat16()
    x = torch.randn(256, 39, 64, device=dev).bfloat16()
    calls = []
    orig = F_.fused_mlp
    F_.fused_mlp = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        xa = x.clone().requires_grad_()
        ya = lay(xa)
    finally:
        F_.fused_mlp = orig
    assert calls, "the layer did not take the fused path"
    assert ya.names == ("B", "N", "O")
    go = torch.randn_like(ya.rename(None))
    ya.rename(None).backward(go)
    ga = [p.grad.clone() for p in lay.parameters()]
    lay.zero_grad()
    saved = F_.FUSED_MLP
    F_.FUSED_MLP = False
    try:
        xb = x.clone().requires_grad_()
        yb = lay(xb)
    finally:
        F_.FUSED_MLP = saved
    yb.rename(None).backward(go)
    # two bf16 paths against EACH OTHER: each is within north_star's 1e-2 of the fp32 oracle (asserted above and in
    # test_gpu_mlp.py), so they may be up to 2e-2 apart (measured: 2.4e-3 forward, 9.7e-3 on the input gradient)
    assert rel_err(ya.rename(None).float(), yb.rename(None).float()) <= 1e-2
    assert rel_err(xa.grad.float(), xb.grad.float()) <= 2e-2
    for a, p in zip(ga, lay.parameters()):
        assert rel_err(a.float(), p.grad.float()) <= 1e-2


def test_mlp_layer_with_wide_first_layer_takes_hybrid_path(dev):
    """DeepFM's deep branch (2496 -> 400 -> 400 -> 400 -> 1 on B rows): first layer on hipBLASLt, everything behind it
    in the fused kernels, one autograd node (layers._HybridMLP: the first layer's ReLU-backward and bias gradient come out
    of the fused backward kernel).  Checked against the fp32 oracle on the forward, and against the all-GEMM
    arrangement it replaces on every gradient (two bf16 paths, each within 1e-2 of the oracle: up to 2e-2 apart)."""
    from torecsys_amd import functional as F_
    from torecsys_amd import layers as L
    torch.manual_seed(7)
    lay = L.DNNLayer(inputs_size=2496, output_size=1, layer_sizes=[400, 400, 400]).to(dev).bfloat16()
    x = (0.5 * torch.randn(8192 + 37, 2496, device=dev)).bfloat16()          # ragged last row tile
    calls = []
    orig = L._HybridMLP.apply
    L._HybridMLP.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        xa = x.clone().requires_grad_()
        ya = lay(xa)
    finally:
        L._HybridMLP.apply = orig
    assert calls, "the layer did not take the hybrid path"
    assert ya.names == ("B", "O") and ya.shape == (x.shape[0], 1)
    lin = [m for m in lay.model if isinstance(m, torch.nn.Linear)]
    yo = O.mlp(x.float().cpu(), [m.weight.detach().float().cpu() for m in lin], [m.bias.detach().float().cpu() for m in lin])
    assert rel_err(ya.rename(None).float().cpu(), yo) <= TOL
    go = torch.randn_like(ya.rename(None))
    ya.rename(None).backward(go)
    ga = [p.grad.clone() for p in lay.parameters()]
    lay.zero_grad()
    saved = L.HYBRID_MLP
    L.HYBRID_MLP = False
    try:
        xb = x.clone().requires_grad_()
        yb = lay(xb)
    finally:
        L.HYBRID_MLP = saved
    yb.rename(None).backward(go)
    assert rel_err(ya.rename(None).float(), yb.rename(None).float()) <= 2 * TOL
    assert rel_err(xa.grad.float(), xb.grad.float()) <= 2 * TOL
    for (n, p), g in zip(lay.named_parameters(), ga):
        assert g.shape == p.grad.shape
        assert rel_err(g.float(), p.grad.float()) <= 2 * TOL, n


@pytest.mark.parametrize("rows,out_f,in_f,x_stride", [(65536, 400, 2496, 512), (1000, 400, 2496, 512), (129, 72, 200, 96),
                                                      (4133, 512, 1024, 512), (257, 8, 40, 32), (128, 33 * 8, 2504, 288)])
def test_rows_gemm_is_the_input_gradient_of_a_linear(dev, rows, out_f, in_f, x_stride):
    """trs_rows_gemm: y = x[:, :out_f] @ W for short K / wide N (ragged last row tile, widths that are not multiples of 32,
    columns of x past out_f that must not leak in) against the same product in fp32 on the bf16-rounded operands"""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(rows + out_f)
    x = torch.randn(rows, x_stride, generator=g).bfloat16()          # garbage past out_f: only zero weight rows may meet it
    x[:, out_f:(out_f + 31) // 32 * 32] = 0          # the contract: readable, and multiplied by zero rows of W's padding
    W = (torch.randn(out_f, in_f, generator=g) / out_f ** 0.5).bfloat16()
    y = F_.rows_gemm(x.to(dev), W.to(dev), out_f, in_f)
    ref = x[:, :out_f].float() @ W.float()
    assert y.shape == (rows, in_f) and y.dtype == torch.bfloat16
    assert rel_err(y.float().cpu(), ref) <= TOL
    assert rel_err_rows(y.float().cpu(), ref, floor_frac=5e-2) <= 2 * TOL


@pytest.mark.parametrize("rows,M,N", [
    (8192, 416, 416),        # 2 x 2 blocks of 13 x 13 tiles, waves 7|6 x 7|6
    (4096 + 40, 416, 416),   # rows not a multiple of 128: the checked loads, zero rows past the end
    (16384, 416, 64),        # narrow x: one workgroup owns all 26 x 4 tiles (waves 4 x 1)
    (16384, 64, 416),        # narrow g (waves 1 x 4)
    (16384, 8, 416),         # the last layer of a stack (one output, padded to 8 columns): half a tile of g
    (9000, 232, 136),        # 15 x 9 tiles: blocks of 7|8 x 9 tiles, waves with 3 and 4 rows
    (5000, 48, 72),
    (4096, 416, 2496),       # 2 x 16 blocks
])
def test_wgrad_rows_kernel_against_fp32_product(dev, rows, M, N):
    """trs_wgrad_rows + trs_wgrad_finish: dW = g^T x summed over the rows, against the fp32 product of the bf16-rounded
    operands (transpose-detecting: M != N in most cases, random operands).  fp32 accumulation in a different order
    than the reference: 1e-5 of the largest entry."""
    from torecsys_amd import functional as F_
    gen = torch.Generator().manual_seed(rows + M)
    g = torch.randn(rows, M, generator=gen).bfloat16()
    x = torch.randn(rows, N, generator=gen).bfloat16()
    S = int(F_._abi.load().trs_wgrad_rows_splits(M, N, rows))
    assert S > 0 and S % 8 == 0
    gw = F_._wgrad_rows(g.to(dev), x.to(dev), M, N, torch.float32)
    ref = g.double().t() @ x.double()
    assert gw.shape == (M, N)
    assert float((gw.double().cpu() - ref).abs().max() / ref.abs().max()) <= 1e-5


def test_wgrad_rows_kernel_honours_row_strides_and_corner(dev):
    """operands that are column slices of wider tensors (ld > M, N) and an output corner smaller than the padded widths
    (what the MLP stack asks for: 400 x 400 out of 416 x 416)"""
    from torecsys_amd import functional as F_
    from torecsys_amd.functional import call, ptr, stream_ptr, _abi
    rows, ld = 4096, 416
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(rows, ld, generator=gen).bfloat16().to(dev)
    x = torch.randn(rows, ld, generator=gen).bfloat16().to(dev)
    M, N = 208, 96
    S = int(_abi.load().trs_wgrad_rows_splits(M, N, rows))
    part = torch.empty(S, M, N, dtype=torch.float32, device=dev)
    call("trs_wgrad_rows", ptr(g), ld, ptr(x), ld, rows, M, N, _abi.TRS_BF16, S, ptr(part), stream_ptr())
    ref = g[:, :M].double().t() @ x[:, :N].double()
    assert float((part.sum(0).double() - ref).abs().max() / ref.abs().max()) <= 1e-5
    gw = F_._wgrad_rows(g, x, 400, 400, torch.bfloat16)
    ref2 = (g.double().t() @ x.double())[:400, :400]
    assert gw.shape == (400, 400) and rel_err(gw.float().cpu(), ref2.float().cpu()) <= TOL


def test_wgrad_rows_falls_back_to_the_library_gemm_below_256_rows(dev):
    from torecsys_amd import functional as F_
    assert int(F_._abi.load().trs_wgrad_rows_splits(416, 416, 200)) == 0
    g = torch.randn(200, 48).bfloat16().to(dev)
    x = torch.randn(200, 72).bfloat16().to(dev)
    gw = F_._wgrad_rows(g, x, 40, 70, torch.float32)
    ref = (g.double().t() @ x.double())[:40, :70]
    assert float((gw.double() - ref).abs().max() / ref.abs().max()) <= 1e-2


@pytest.mark.parametrize("rows", [1000 + 19, 66000 + 7])
def test_fused_backward_masks_its_input_gradient_with_the_upstream_relu(dev, rows, ro_mode):
    """mask_in / gbias_in of trs_mlp_fused_*: for a stack fed by relu(z), gx must be dL/dz = dL/dx * [x > 0] and gbias_in
    its column sums -- against the unmasked gradient of the same call, masked and summed here (ragged last row tile,
    input columns that are exactly zero in some rows)"""
    from torecsys_amd import functional as F_
    gen = torch.Generator().manual_seed(11)
    widths = [416, 400, 400, 8]
    x = torch.relu(torch.randn(rows, widths[0], generator=gen)).bfloat16().to(dev)
    Ws = [(torch.randn(widths[l + 1], widths[l], generator=gen) / widths[l] ** 0.5).bfloat16().to(dev) for l in range(3)]
    bs = [(0.1 * torch.randn(widths[l + 1], generator=gen)).bfloat16().to(dev) for l in range(3)]
    y, hidden, masks, mask_in, fam = F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=True)
    y0, _, _, fam0 = F_.fused_mlp_forward_raw(x, Ws, bs)
    assert torch.equal(y, y0) and fam == fam0 == ro_mode
    gy = torch.randn(rows, widths[-1], generator=gen).bfloat16().to(dev)
    gx0, gz0, gb0, none = F_.fused_mlp_backward_raw(gy, widths, Ws, masks, family=fam)
    gx1, gz1, gb1, gb_in = F_.fused_mlp_backward_raw(gy, widths, Ws, masks, mask_in, family=fam)
    assert none is None
    assert all(torch.equal(a, b) for a, b in zip(gz0, gz1)) and all(torch.equal(a, b) for a, b in zip(gb0, gb1))
    want = torch.where(x > 0, gx0, torch.zeros_like(gx0))
    assert torch.equal(gx1, want)
    ref = want.double().sum(0)
    assert float((gb_in[:widths[0]].double() - ref).abs().max() / ref.abs().max()) <= 1e-5


@pytest.mark.parametrize("fwd_req,bwd_req", [(2, 1), (1, 2), (2, 0), (0, 2)])
def test_backward_runs_the_family_its_forward_ran_whatever_is_requested_later(dev, fwd_req, bwd_req, monkeypatch):
    """The kernel family is per call and the autograd node records it: a request (or a size policy) that changes between
    a stack's forward and its backward must not change which backward kernels read the forward's sign bits.  Forward under
    one request, backward under another: gradients equal the ones of a run that never switched, bit for bit.  (Round 4
    had a process-global mode here; flipping it between the two calls gave wrong gradients with return code 0.)"""
    from torecsys_amd import functional as F_
    monkeypatch.setattr(F_, "FUSED_MLP_MIN_ROWS", 1)
    g = torch.Generator().manual_seed(77)
    widths = [64, 400, 400, 400, 64]
    Ws, bs = _params(widths, g)
    x = torch.randn(3000, 64, generator=g).bfloat16().to(dev)
    gy = torch.randn(3000, 64, generator=g).bfloat16().to(dev)

    def run(req_f, req_b):
        xd = x.clone().requires_grad_()
        Wd = [w.to(dev).requires_grad_() for w in Ws]
        bd = [b.to(dev).requires_grad_() for b in bs]
        with F_.mlp_family(req_f):
            y = F_.fused_mlp(xd, Wd, bd)
        with F_.mlp_family(req_b):
            y.backward(gy)
        return [y.detach(), xd.grad] + [w.grad for w in Wd] + [b.grad for b in bd]

    same = run(fwd_req, fwd_req)
    switched = run(fwd_req, bwd_req)
    assert all(torch.equal(a, b) for a, b in zip(same, switched))


def test_backward_refuses_a_family_the_forward_did_not_report(dev):
    """C ABI: trs_mlp_fused_bwd_data takes the family as an argument -- AUTO (a policy, not a record) and values outside
    {TILE, ROW_OWNER} are TRS_EINVAL, and so is ROW_OWNER for a stack those kernels do not cover; the Python wrapper
    raises before the call."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(3)
    widths = [32, 104, 200, 40]              # not a row-owner shape
    Ws, bs = _params(widths, g)
    Ws, bs = [w.to(dev) for w in Ws], [b.to(dev) for b in bs]
    x = torch.randn(512, 32, generator=g).bfloat16().to(dev)
    y, hidden, masks, fam = F_.fused_mlp_forward_raw(x, Ws, bs, family=F_.MLP_FAMILY_ROW_OWNER)   # falls back: uncovered
    assert fam == F_.MLP_FAMILY_TILE
    assert int(F_._abi.load().trs_mlp_fused_family(3, F_._i32_array(widths), 512, F_.MLP_FAMILY_ROW_OWNER)) == 0
    gy = torch.randn(512, 40, generator=g).bfloat16().to(dev)
    with pytest.raises(ValueError):
        F_.fused_mlp_backward_raw(gy, widths, Ws, masks, family=F_.MLP_FAMILY_AUTO)
    gz = [torch.empty(512, F_._pad32(w), dtype=torch.bfloat16, device=dev) for w in widths[1:-1]]
    gb = [torch.empty(F_._pad32(w), dtype=torch.float32, device=dev) for w in widths[1:]]
    gx = torch.empty(512, 32, dtype=torch.bfloat16, device=dev)
    wl = F_._i32_array(widths)
    ws_bytes = F_.size_query("trs_mlp_fused_workspace_bytes", 3, wl)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    for bad in (F_.MLP_FAMILY_AUTO, F_.MLP_FAMILY_ROW_OWNER, 7):
        with pytest.raises(RuntimeError, match="family"):
            F_.call("trs_mlp_fused_bwd_data", F_.ptr(gy), 512, 3, wl, F_._ptr_array(Ws), F_._ptr_array(masks),
                    F_._ptr_array(gz), F_._ptr_array(gb), F_.ptr(gx), F_.ptr(None), F_.ptr(None), F_._abi.TRS_BF16, bad,
                    F_.MLP_PHASE_ALL, F_.ptr(ws), ws_bytes, F_.stream_ptr())


@pytest.mark.parametrize("widths,rows", [([32, 104, 200, 40], 700), ([416, 400, 400, 8], 3000)])
@pytest.mark.parametrize("fam_req", [1, 2, 3])
def test_pack_and_run_phases_equal_the_one_call_form(dev, widths, rows, fam_req):
    """C ABI phases: a PACK call (weights into fragment order, on a SIDE stream) followed by a RUN call on the packed
    workspace gives bit for bit what the one-call form gives -- forward, data gradient, bias gradients, and trs_rows_gemm."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(5)
    Ws, bs = _params(widths, g)
    Ws, bs = [w.to(dev) for w in Ws], [b.to(dev) for b in bs]
    x = torch.randn(rows, widths[0], generator=g).relu().bfloat16().to(dev)
    gy = torch.randn(rows, widths[-1], generator=g).bfloat16().to(dev)
    fam = F_.mlp_fused_family(widths, rows, fam_req)
    y0, h0, m0, mi0, _ = F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=True, family=fam)
    ref_b = F_.fused_mlp_backward_raw(gy, widths, Ws, m0, mi0, family=fam)
    (wsf, wsb), ev, side = F_.run_on_side(dev, "pack", lambda: (F_.fused_mlp_pack(Ws, bs, widths, rows, fam, False),
                                                                F_.fused_mlp_pack(Ws, None, widths, rows, fam, True)))
    torch.cuda.current_stream().wait_event(ev)
    y1, h1, m1, mi1, _ = F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=True, family=fam, packed_ws=wsf)
    got_b = F_.fused_mlp_backward_raw(gy, widths, Ws, m1, mi1, family=fam, packed_ws=wsb)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and all(torch.equal(a, b) for a, b in zip(h0, h1))
    assert torch.equal(ref_b[0], got_b[0]) and all(torch.equal(a, b) for a, b in zip(ref_b[1], got_b[1]))
    for a, b in zip(list(ref_b[2]) + [ref_b[3]], list(got_b[2]) + [got_b[3]]):
        assert torch.equal(a, b)
    # the wide input gradient in front of such a stack
    out_f, in_f = widths[0] - (16 if widths[0] == 416 else 0), 1024
    W1 = (torch.randn(out_f, in_f, generator=g) * 0.05).bfloat16().to(dev)
    gz = torch.randn(4096, widths[0], generator=g).bfloat16().to(dev)
    if F_.rows_gemm_supported(gz, W1, out_f, in_f):
        ws = F_.rows_gemm_pack(W1, 4096, widths[0], out_f, in_f)
        assert torch.equal(F_.rows_gemm(gz, W1, out_f, in_f), F_.rows_gemm(gz, W1, out_f, in_f, packed_ws=ws))


def test_mixed_family_reads_the_first_columns_of_wider_rows(dev):
    """The deep branch's arrangement: the library GEMM in front of the fused tail leaves 512-wide rows (zero beyond the 400
    real columns), the tail's FORWARD reads their first 416 columns (x_stride) with the row-owner kernel, its BACKWARD runs
    the tile kernels at the full 512 columns on the row-owner sign bits.  Equal to the 416-wide stack on the 416-wide copy
    of the rows: output, hidden activations, input gradient (zeros in columns 416..511), inner gradients, bias gradients."""
    from torecsys_amd import functional as F_
    g = torch.Generator().manual_seed(9)
    rows, wn, ww = 40000, 416, 512
    widths = [wn, 400, 400, 8]
    Ws, bs = _params([400, 400, 400, 8], g)
    W0n = torch.zeros(400, wn, dtype=torch.bfloat16); W0n[:, :400] = Ws[0]
    W0w = torch.zeros(400, ww, dtype=torch.bfloat16); W0w[:, :400] = Ws[0]
    Wn = [W0n.to(dev)] + [w.to(dev) for w in Ws[1:]]
    Ww = [W0w.to(dev)] + [w.to(dev) for w in Ws[1:]]
    bd = [b.to(dev) for b in bs]
    xw = torch.zeros(rows, ww, dtype=torch.bfloat16)
    xw[:, :400] = torch.randn(rows, 400, generator=g).relu().bfloat16()
    xw = xw.to(dev)
    xn = xw[:, :wn].contiguous()
    gy = torch.randn(rows, 8, generator=g).bfloat16().to(dev)
    assert F_.mlp_fused_family(widths, rows, F_.MLP_FAMILY_MIXED) == F_.MLP_FAMILY_MIXED
    y0, h0, m0, mi0, fam0 = F_.fused_mlp_forward_raw(xn, Wn, bd, input_mask=True, family=F_.MLP_FAMILY_MIXED)
    y1, h1, m1, mi1, fam1 = F_.fused_mlp_forward_raw(xw, Wn, bd, input_mask=True, family=F_.MLP_FAMILY_MIXED, x_stride=ww)
    assert fam0 == fam1 == F_.MLP_FAMILY_MIXED
    assert torch.equal(y0, y1) and all(torch.equal(a, b) for a, b in zip(h0, h1))
    gx0, gz0, gb0, gbi0 = F_.fused_mlp_backward_raw(gy, widths, Wn, m0, mi0, family=fam0)
    gx1, gz1, gb1, gbi1 = F_.fused_mlp_backward_raw(gy, [ww] + widths[1:], Ww, m1, mi1, family=fam1)
    torch.cuda.synchronize()
    assert gx1.shape == (rows, ww) and torch.equal(gx1[:, :wn], gx0) and not gx1[:, wn:].any()
    assert all(torch.equal(a, b) for a, b in zip(gz0, gz1))
    for a, b in zip(gb0, gb1):
        assert rel_err(b.cpu(), a.cpu()) <= 1e-6
    assert rel_err(gbi1[:wn].cpu(), gbi0.cpu()) <= 1e-6 and not gbi1[wn:].any()
    # the tile kernels do not read wider rows
    with pytest.raises(RuntimeError, match="x_stride"):
        F_.fused_mlp_forward_raw(xw, Wn, bd, family=F_.MLP_FAMILY_TILE, x_stride=ww)


@pytest.mark.parametrize("M,N", [(416, 416), (400, 400)])
def test_wgrad_rows_long_row_ranges_take_the_four_wave_dma_kernel(dev, M, N):
    """From 2^20 rows on a 13|12 x 26|25-tile weight gradient runs four waves on 13 x 26-tile blocks fed by LDS-DMA
    (wgrad_dma2_kernel: twice the row ranges of the eight-wave form); against the float64 product on the device."""
    from torecsys_amd import functional as F_
    rows, ld = (1 << 20) + 128 * 3, 416
    gen = torch.Generator(device=dev).manual_seed(M)
    g = torch.randn(rows, ld, generator=gen, device=dev).bfloat16()
    x = torch.randn(rows, ld, generator=gen, device=dev).bfloat16()
    S8 = int(F_._abi.load().trs_wgrad_rows_splits(M, N, 131072))      # the eight-wave form
    S = int(F_._abi.load().trs_wgrad_rows_splits(M, N, rows))
    assert S == 2 * S8, (S, S8)
    gw = F_._wgrad_rows(g, x, M, N, torch.float32)
    ref = torch.zeros(M, N, dtype=torch.float64, device=dev)
    for r0 in range(0, rows, 1 << 17):
        ref += g[r0:r0 + (1 << 17), :M].double().t() @ x[r0:r0 + (1 << 17), :N].double()
    assert gw.shape == (M, N)
    assert float((gw.double() - ref).abs().max() / ref.abs().max()) <= 1e-5


@pytest.mark.parametrize("div", [2, 4, 64])
def test_wgrad_rows_with_fewer_row_ranges_than_planned(dev, div):
    """trs_wgrad_rows accepts the planned number of row ranges halved down to 8 (longer ranges on fewer workgroups:
    layers.WGRAD_PAIR runs two such launches side by side); any other count is refused"""
    from torecsys_amd import functional as F_
    from torecsys_amd.functional import call, ptr, stream_ptr, _abi
    rows, M, N = 65536, 400, 400
    gen = torch.Generator().manual_seed(div)
    g = torch.randn(rows, 416, generator=gen).bfloat16().to(dev)
    x = torch.randn(rows, 416, generator=gen).bfloat16().to(dev)
    S0 = int(_abi.load().trs_wgrad_rows_splits(M, N, rows))
    assert S0 >= 16
    gw = F_._wgrad_rows(g, x, M, N, torch.float32, splits_div=div)
    ref = (g.double().t() @ x.double())[:M, :N]
    assert float((gw.double() - ref).abs().max() / ref.abs().max()) <= 1e-5
    part = torch.empty(S0, M, N, dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError):
        call("trs_wgrad_rows", ptr(g), 416, ptr(x), 416, rows, M, N, _abi.TRS_BF16, 24, ptr(part), stream_ptr())
    with pytest.raises(RuntimeError):
        call("trs_wgrad_rows", ptr(g), 416, ptr(x), 416, rows, M, N, _abi.TRS_BF16, 2 * S0, ptr(part), stream_ptr())


def test_hybrid_branch_with_paired_tail_weight_gradients(dev):
    """layers.WGRAD_PAIR: the two square tail layers' weight gradients as half-size launches on two streams -- the same
    gradients up to the fp32 summation order of the row ranges, eagerly and replayed from a hipGraph"""
    from torecsys_amd import layers as L
    from torecsys_amd.graph import GraphedStep
    torch.manual_seed(11)
    lay = L.DNNLayer(inputs_size=2496, output_size=1, layer_sizes=[400, 400, 400]).to(dev).bfloat16()
    params = list(lay.parameters())
    x = (0.5 * torch.randn(16384, 2496, device=dev)).bfloat16()
    go = torch.randn(16384, 1, device=dev).bfloat16()

    def run(xin):
        xa = xin.clone().requires_grad_()
        (lay(xa).rename(None) * go).sum().backward()
        return xa.grad

    def fresh():
        for p in params:
            p.grad = None

    fresh()
    gx0 = run(x)
    g0 = [p.grad.clone() for p in params]
    saved = L.WGRAD_PAIR
    L.WGRAD_PAIR = True
    try:
        fresh()
        gx1 = run(x)
        torch.cuda.synchronize()
        assert torch.equal(gx0, gx1)
        for a, p in zip(g0, params):
            assert rel_err(a.float(), p.grad.float()) <= 1e-3
        g1 = [p.grad.clone() for p in params]
        fresh()
        xs = x.clone()
        step = GraphedStep(lambda xin: run(xin).float().sum(), (xs,), params=params, warmup=2)
        for _ in range(3):
            step(xs)
        torch.cuda.synchronize()
        for a, p in zip(g1, params):
            assert torch.equal(a, p.grad)
    finally:
        L.WGRAD_PAIR = saved
