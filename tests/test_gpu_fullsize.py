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


# ------------------------------------------------------------------------------------------------------------------------
# The three timed models, the step exactly as bench.py arranges it (Inputs router over the fused lookup + FM table and the
# E = 1 first-order table, the harness model, the fused head, the HIP BCE-with-logits loss, full backward with dense table
# gradients; DeepFM replayed from a hipGraph, DCN / xDeepFM eager), at B = 65 536 on a 1 M-row table.
#
# How a sum over 65 536 samples is checked against an oracle that can only afford a few dozen of them: the S sampled
# samples look up RESERVED table rows (sample j reads row j of every field; every other sample draws from rows >= S), so
# the gradient rows of those reserved ids receive exactly one contribution each -- (1/B) dl_j / d emb[j, n] -- and the
# oracle evaluated on the S samples alone, with its loss divided by B, must reproduce them.
# ------------------------------------------------------------------------------------------------------------------------
V_STEP = 1_000_000
# Per-SAMPLE norm of a table gradient that passed through a ReLU stack the oracle cannot replay under the kernel's own
# masks (the [400,400,400] MLPs; CIN's masks ARE replayed): a hidden unit whose pre-activation lies within the bf16
# pipeline's error of zero (relative 2^-8 .. 2^-7 of the layer's scale: ~1 % of the 1 200 units of a sample) may take the
# other side of the ReLU than in the fp32 oracle; each such unit moves the sample's gradient by its own share of it,
# ~1/sqrt(400) = 5 % of the sample's largest entry at worst, a handful of them adding in quadrature.  Bounded at 5e-2 per
# sample, while the max norm over all sampled rows (rel_err: where a flipped unit cannot hide either) stays at TOL.
TOL_ROWS_RELU = 5e-2


def _step_case(dev, seed):
    per = V_STEP // N
    sizes = [per] * (N - 1) + [V_STEP - per * (N - 1)]
    off = O.field_offsets(sizes)
    g = torch.Generator().manual_seed(seed)
    idx = torch.cat([torch.randint(S, f, (B, 1), generator=g) for f in sizes], 1)
    rows = torch.randperm(B, generator=g)[:S].sort().values
    idx[rows] = torch.arange(S).view(S, 1).expand(S, N)
    labels = (torch.rand(B, 1, generator=g) < 0.25).float()
    return sizes, off, idx, rows, labels


def _step_inputs(dev, sizes):
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    emb = MultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=True)
    feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
    emb.set_schema(["c0"])
    feat.set_schema(["c0"])
    return Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev).to(torch.bfloat16), emb, feat


def _reserved(emb, feat, off, dev):
    """fp32 host copies of the reserved rows of both tables as (S*N, .) leaf tensors + the (S, N) local index block"""
    gid = (torch.arange(S).view(S, 1) + off.view(1, N)).reshape(-1)
    w = emb.embedding.weight.detach()[gid.to(dev)].float().cpu().requires_grad_()
    w1 = feat.embedding.weight.detach()[gid.to(dev)].float().cpu().requires_grad_()
    return gid, w, w1, torch.arange(S * N).view(S, N)


def _mlp_of(dnn):
    lin = [m for m in dnn.model if isinstance(m, torch.nn.Linear)]
    return [l.weight.detach().float().cpu() for l in lin], [l.bias.detach().float().cpu() for l in lin]


def _check_step(tag, logits, ref_logits, terms, gw, gw_ref, gw1, gw1_ref, loss, loss_ref):
    le = sum_err(logits, ref_logits, terms)
    ge, gr = rel_err(gw, gw_ref), rel_err_rows(gw.view(S, -1), gw_ref.view(S, -1))
    g1 = rel_err(gw1, gw1_ref)
    print(f"{tag}: logits sum_err {le:.2e}  table-grad rel {ge:.2e} per-sample-row {gr:.2e}  first-order grad {g1:.2e}"
          + (f"  loss {loss:.6f} vs {loss_ref:.6f}" if loss_ref is not None else ""))
    return le, ge, gr, g1


def test_deepfm_step_full_size(dev):
    """DeepFM (deep_fm.py:73-108) as benchmarked: graph replay, mixed-family fused tail, fused head, BCE.  Sampled-row
    logits, the reserved rows of BOTH table gradients, and -- the oracle affords the whole batch for this model -- the
    loss itself over all 65 536 samples."""
    from harness import ctr_models as M
    from torecsys_amd.fused import BCEWithLogitsLoss
    from torecsys_amd.graph import GraphedStep
    sizes, off, idx, rows, labels = _step_case(dev, 2024)
    torch.manual_seed(7)
    inputs, emb, feat = _step_inputs(dev, sizes)
    model = M.DeepFactorizationMachineModel(embed_size=E, num_fields=N, deep_layer_sizes=[400, 400, 400],
                                            fm_dropout_p=0.0).to(dev).to(torch.bfloat16)
    crit = BCEWithLogitsLoss()
    params = [p for p in list(inputs.parameters()) + list(model.parameters()) if p.requires_grad]
    held = {}

    def fn(ix, lab):
        d = inputs({"c0": ix})
        out = model(**d)
        held["out"] = out.detach()
        l_ = crit(out, lab)
        l_.backward()
        return l_

    step = GraphedStep(fn, (idx.to(dev), labels.to(dev)), params=params, warmup=1)
    other = torch.cat([torch.randint(S, f, (B, 1), generator=torch.Generator().manual_seed(1)) for f in sizes], 1)
    step(other.to(dev), labels.to(dev))          # another batch through the same graph first: nothing may be stale
    loss = step(idx.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    logits = held["out"].float().cpu()
    ws, bs = _mlp_of(model.deep)
    # the whole batch on the oracle (fp32 on the bf16-rounded parameters): loss and every logit
    W = emb.embedding.weight.detach().float().cpu()
    W1 = feat.embedding.weight.detach().float().cpu()
    e_all = O.multi_indices_embedding(W, idx, off)
    f_all = O.multi_indices_embedding(W1, idx, off)
    ref_all = O.deepfm_model(f_all, e_all, ws, bs)
    loss_ref = float(O.bce_with_logits(ref_all, labels))
    assert abs(float(loss) - loss_ref) <= 1e-3 * abs(loss_ref)
    assert rel_err(logits, ref_all) <= TOL
    gid, w, w1, loc = _reserved(emb, feat, off, dev)
    er = O.multi_indices_embedding(w, loc, torch.zeros(N, dtype=torch.int64))
    fr = O.multi_indices_embedding(w1, loc, torch.zeros(N, dtype=torch.int64))
    ref = O.deepfm_model(fr, er, ws, bs)
    (torch.nn.functional.binary_cross_entropy_with_logits(ref, labels[rows], reduction="sum") / B).backward()
    with torch.no_grad():       # the logit is sum_e FM_e + sum_n first_n + deep: bounded relative to its terms
        terms = (O.fm_layer(er).abs().sum(dim=1, keepdim=True) + fr.abs().sum(dim=1)
                 + O.mlp(er.reshape(S, -1), ws, bs).abs())
    le, ge, gr, g1 = _check_step("deepfm", logits[rows], ref.detach(), terms,
                                 emb.embedding.weight.grad[gid.to(dev)].float().cpu(), w.grad,
                                 feat.embedding.weight.grad[gid.to(dev)].float().cpu(), w1.grad, float(loss), loss_ref)
    assert le <= TOL and ge <= TOL and gr <= TOL and g1 <= TOL      # (measured 6e-3 per sample: no looser bound needed here)


def test_dcn_step_full_size(dev):
    """DeepAndCrossNetwork (deep_and_cross_network.py:71-96) as benchmarked: cross network (6 layers, detached first
    input) + row-owner per-field MLP at 2.56 M rows + the cat-free head + BCE, eager."""
    from harness import ctr_models as M
    from torecsys_amd.fused import BCEWithLogitsLoss
    sizes, off, idx, rows, labels = _step_case(dev, 2025)
    torch.manual_seed(11)
    inputs, emb, feat = _step_inputs(dev, sizes)
    model = M.DeepAndCrossNetworkModel(inputs_size=E, num_fields=N, deep_output_size=64, deep_layer_sizes=[400, 400, 400],
                                       cross_num_layers=6).to(dev).to(torch.bfloat16)
    crit = BCEWithLogitsLoss()
    for _ in range(2):
        for p in list(inputs.parameters()) + list(model.parameters()):
            p.grad = None
        d = inputs({"c0": idx.to(dev)})
        out = model(emb_inputs=d["emb_inputs"])
        loss = crit(out, labels.to(dev))
        loss.backward()
    torch.cuda.synchronize()
    logits = out.detach().float().cpu()
    gid, w, _, loc = _reserved(emb, feat, off, dev)
    er = O.multi_indices_embedding(w, loc, torch.zeros(N, dtype=torch.int64))
    f32 = lambda t: t.detach().float().cpu()
    cw, cb = [f32(l.weight) for l in model.cross.model], [f32(l.bias) for l in model.cross.model]
    ws, bs = _mlp_of(model.deep)
    ref = O.dcn_model(er, cw, cb, ws, bs, f32(model.fc.weight), f32(model.fc.bias))
    (torch.nn.functional.binary_cross_entropy_with_logits(ref, labels[rows], reduction="sum") / B).backward()
    # the logit is one 4 992-term dot product of [cross | deep] with the head's weights: bounded relative to its terms
    with torch.no_grad():
        cat = torch.cat([O.cross_network(er, cw, cb), O.mlp(er, ws, bs)], dim=2).reshape(S, -1)
        terms = cat.abs() @ f32(model.fc.weight).abs().t() + f32(model.fc.bias).abs()
    gw = emb.embedding.weight.grad[gid.to(dev)].float().cpu()
    le = sum_err(logits[rows], ref.detach(), terms)
    ge, gr = rel_err(gw, w.grad), rel_err_rows(gw.view(S, -1), w.grad.view(S, -1))
    print(f"dcn: logits sum_err {le:.2e}  table-grad rel {ge:.2e} per-sample-row {gr:.2e}  loss {float(loss):.6f}")
    assert le <= TOL and ge <= TOL and gr <= TOL_ROWS_RELU


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_xdeepfm_step_full_size(dev, mode):
    """xDeepFM (xdeep_fm.py:100-122) as benchmarked: CIN [128,128,128] + DNN + first-order sum + bias, fused head, BCE.
    train: BatchNorm on batch statistics -- reduced in float64 on the device from each layer's own contraction output and
    handed to the oracle -- sampled-row LOGITS (the train-mode backward couples every sample through the statistics; its
    kernels are pinned at this batch size in test_cin_pieces_full_size).  eval: running statistics, samples independent:
    logits and the reserved table-gradient rows, the oracle's CIN under the kernel's own ReLU masks."""
    from harness import ctr_models as M
    from test_gpu_cin_parity import _GlueRecorder
    from torecsys_amd import functional as F_
    from torecsys_amd.fused import BCEWithLogitsLoss
    sizes, off, idx, rows, labels = _step_case(dev, 2026)
    rd = rows.to(dev)
    torch.manual_seed(12)
    inputs, emb, feat = _step_inputs(dev, sizes)
    model = M.XDeepFactorizationMachineModel(embed_size=E, num_fields=N, cin_layer_sizes=[128, 128, 128],
                                             deep_layer_sizes=[400, 400, 400])
    for seq in model.cin.model:
        seq.Batchnorm.running_mean.normal_(0.0, 0.05)
        seq.Batchnorm.running_var.uniform_(0.5, 1.5)
        seq.Batchnorm.weight.data.uniform_(0.5, 1.5)
        seq.Batchnorm.bias.data.normal_(0.0, 0.2)
    model = model.to(dev).to(torch.bfloat16)
    model.train(mode == "train")
    crit = BCEWithLogitsLoss()
    stats_before = [(seq.Batchnorm.running_mean.clone(), seq.Batchnorm.running_var.clone()) for seq in model.cin.model]
    d = inputs({"c0": idx.to(dev)})
    with _GlueRecorder(F_) as rec:
        out = model(**d)
    loss = crit(out, labels.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    assert len(rec.seen) == 3
    logits = out.detach().float().cpu()
    P = _cin_params(model.cin)
    stats, masks = [], []
    for k, (yT, bn, D, Hs) in enumerate(rec.seen):
        if mode == "train":
            y64 = yT.double()
            mean, var = y64.mean(dim=(0, 1)), y64.var(dim=(0, 1), unbiased=False)
            del y64
        else:
            mean, var = stats_before[k][0].double(), stats_before[k][1].double()
        scale, shift = _affine_of(yT, bn, mean, var)
        z = yT[rd].float() * scale.float() + shift.float()
        masks.append((z > 0).float().transpose(1, 2).contiguous().cpu())
        stats.append((mean.float().cpu(), var.float().cpu()))
    gid, w, w1, loc = _reserved(emb, feat, off, dev)
    zero_off = torch.zeros(N, dtype=torch.int64)
    er, fr = O.multi_indices_embedding(w, loc, zero_off), O.multi_indices_embedding(w1, loc, zero_off)
    ws, bs = _mlp_of(model.deep)
    kw = dict(P, bn_running_means=[m.clone() for m, _ in stats], bn_running_vars=[v.clone() for _, v in stats],
              training=False)
    bias = model.bias.detach().float().cpu()
    ref_plain = O.xdeepfm_model(fr, er, kw, ws, bs, bias)
    with torch.no_grad():
        _, _, pooled = O.cin_layer(er, **kw, return_intermediates=True)
        terms = (pooled.abs() @ P["fc_weight"].detach().abs().t() + P["fc_bias"].abs() + fr.abs().sum(dim=1)
                 + O.mlp(er.reshape(S, -1), ws, bs).abs() + bias.abs())
    le = sum_err(logits[rows], ref_plain.detach(), terms)
    print(f"xdeepfm[{mode}]: logits sum_err {le:.2e}  loss {float(loss):.6f}")
    assert le <= TOL
    if mode == "train":
        return
    acts = [(lambda t, m=m: t * m) for m in masks]
    ref = O.xdeepfm_model(fr, er, dict(kw, activation=acts), ws, bs, bias)
    (torch.nn.functional.binary_cross_entropy_with_logits(ref, labels[rows], reduction="sum") / B).backward()
    gw = emb.embedding.weight.grad[gid.to(dev)].float().cpu()
    gw1 = feat.embedding.weight.grad[gid.to(dev)].float().cpu()
    ge, gr, g1 = rel_err(gw, w.grad), rel_err_rows(gw.view(S, -1), w.grad.view(S, -1)), rel_err(gw1, w1.grad)
    print(f"xdeepfm[eval]: table-grad rel {ge:.2e} per-sample-row {gr:.2e}  first-order grad {g1:.2e}")
    assert ge <= TOL and gr <= TOL_ROWS_RELU and g1 <= TOL
