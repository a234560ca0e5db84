"""Pin the CPU oracle (oracle/cpu_ref.py) against golden vectors captured from the real reference
(tests/golden/make_golden.py).  Gather: bit-exact.  Float results: <=1e-6 relative (fp32)."""
import pytest
import torch

from conftest import CIN_CASES, LAYER_SHAPES, MODEL_SHAPES, rel_err
from oracle import cpu_ref as O

TOL = 1e-6


def _tag(s):
    return "%d_%d_%d" % s


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_multi_indices_embedding(golden, shape):
    G = golden("inputs")
    t = "multi/" + _tag(shape)
    fs = G(t + "/field_sizes").tolist()
    off = O.field_offsets(fs)
    assert torch.equal(off, G(t + "/offsets"))
    assert torch.equal(O.field_offsets(fs, through_float32=True), G(t + "/offsets"))
    w = G(t + "/weight").requires_grad_()
    out = O.multi_indices_embedding(w, G(t + "/idx"), off)
    assert torch.equal(out, G(t + "/out"))                      # bit-exact gather
    assert G(t + "/names") == ["B", "N", "E"]
    (out * G(t + "/gout")).sum().backward()
    assert rel_err(w.grad, G(t + "/gweight")) <= TOL
    flat = O.multi_indices_embedding(w.detach(), G(t + "/idx"), off, flatten=True)
    assert list(flat.shape) == G(t + "/flatten_shape").tolist()
    # E = 1 first-order table
    t1 = "multi1/" + _tag(shape)
    assert torch.equal(O.multi_indices_embedding(G(t1 + "/weight"), G(t + "/idx"), off), G(t1 + "/out"))


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_single_index_embedding(golden, shape):
    G = golden("inputs")
    t = "single/" + _tag(shape)
    w = G(t + "/weight").requires_grad_()
    idx = G(t + "/idx")
    assert idx.dtype == torch.int32
    out = O.single_index_embedding(w, idx, padding_idx=0)
    assert torch.equal(out, G(t + "/out"))
    (out * G(t + "/gout")).sum().backward()
    assert rel_err(w.grad, G(t + "/gweight")) <= TOL
    assert float(w.grad[0].abs().max()) == 0.0                  # padding row gets no gradient


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_field_aware_embedding(golden, shape):
    G = golden("inputs")
    t = "fa/" + _tag(shape)
    tm = "multi/" + _tag(shape)
    ws = [w.clone().requires_grad_() for w in G(t + "/weights")]
    off = O.field_offsets(G(tm + "/field_sizes").tolist())
    out = O.multi_indices_field_aware_embedding(ws, G(tm + "/idx"), off)
    N = shape[1]
    assert out.shape[1] == N * N
    stride = int(G(t + "/out_sub_stride")[0])
    assert torch.equal(out[:, ::stride], G(t + "/out_sub"))
    assert abs(float(out.double().sum()) - float(G(t + "/out_checksum")[0])) <= 1e-9 * out.numel()
    out.sum().backward()
    assert rel_err(ws[1].grad, G(t + "/gweight1_ones")) <= TOL


@pytest.mark.parametrize("shape", LAYER_SHAPES)
def test_fm_ipn_ffm_cross(golden, shape):
    G = golden("layers")
    tag = _tag(shape)
    x = G(f"fm/{tag}/x").requires_grad_()
    y = O.fm_layer(x)
    assert rel_err(y, G(f"fm/{tag}/out")) <= TOL
    assert G(f"fm/{tag}/names") == ["B", "O"]
    (y * G(f"fm/{tag}/gout")).sum().backward()
    assert rel_err(x.grad, G(f"fm/{tag}/gx")) <= TOL

    x = G(f"fm/{tag}/x").requires_grad_()
    y = O.inner_product_layer(x)
    assert rel_err(y, G(f"ipn/{tag}/out")) <= TOL
    assert G(f"ipn/{tag}/names") == ["B", "O"]
    (y * G(f"ipn/{tag}/gout")).sum().backward()
    assert rel_err(x.grad, G(f"ipn/{tag}/gx")) <= TOL

    xf = G(f"ffm/{tag}/x").requires_grad_()
    y = O.ffm_layer(xf, shape[1])
    assert torch.equal(y, G(f"ffm/{tag}/out"))                  # one multiply per element: exact
    assert G(f"ffm/{tag}/names") == ["B", "N", "E"]
    (y * G(f"ffm/{tag}/gout")).sum().backward()
    assert rel_err(xf.grad, G(f"ffm/{tag}/gx")) <= TOL

    x = G(f"cross/{tag}/x").requires_grad_()
    W = [w.clone().requires_grad_() for w in G(f"cross/{tag}/W")]
    b = [v.clone().requires_grad_() for v in G(f"cross/{tag}/b")]
    y = O.cross_network(x, W, b)
    assert rel_err(y, G(f"cross/{tag}/out")) <= TOL
    assert G(f"cross/{tag}/names") == ["B", "N", "O"]
    (y * G(f"cross/{tag}/gout")).sum().backward()
    assert rel_err(x.grad, G(f"cross/{tag}/gx")) <= 5 * TOL
    assert rel_err(torch.stack([w.grad for w in W]), G(f"cross/{tag}/gW")) <= 5 * TOL
    assert rel_err(torch.stack([v.grad for v in b]), G(f"cross/{tag}/gb")) <= 5 * TOL
    # the detach quirk matters: the textbook gradient differs
    x2 = G(f"cross/{tag}/x").requires_grad_()
    y2 = O.cross_network(x2, [w.detach() for w in W], [v.detach() for v in b], detach_first_input=False)
    (y2 * G(f"cross/{tag}/gout")).sum().backward()
    assert rel_err(x2.grad, G(f"cross/{tag}/gx")) > 1e-3


def test_cross_2d_raises_in_reference(golden):
    assert golden("layers")("cross2d/raises") == ["RuntimeError"]


def _cin_kwargs(G, name, grad=False):
    pre = f"cin/{name}"
    B, N, E, direct, use_bias, use_bn = G(pre + "/cfg").tolist()
    L = len(G(pre + "/layer_sizes"))

    def p(k):
        t = G(k).clone()
        return t.requires_grad_() if grad else t
    kw = dict(
        conv_weights=[p(f"{pre}/conv_w{i}") for i in range(L)],
        conv_biases=[p(f"{pre}/conv_b{i}") if use_bias else None for i in range(L)],
        fc_weight=p(pre + "/fc_w"), fc_bias=p(pre + "/fc_b"), is_direct=bool(direct))
    if use_bn:
        kw["bn_weights"] = [p(f"{pre}/bn_w{i}") for i in range(L)]
        kw["bn_biases"] = [p(f"{pre}/bn_b{i}") for i in range(L)]
    return kw, L, bool(use_bias), bool(use_bn)


@pytest.mark.parametrize("name", CIN_CASES)
def test_cin(golden, name):
    G = golden("cin")
    pre = f"cin/{name}"
    kw, L, use_bias, use_bn = _cin_kwargs(G, name, grad=True)
    C = [w.shape[0] for w in kw["conv_weights"]]
    if use_bn:
        kw["bn_running_means"] = [torch.zeros(c) for c in C]
        kw["bn_running_vars"] = [torch.ones(c) for c in C]
    x = G(pre + "/x").requires_grad_()
    y = O.cin_layer(x, training=True, **kw)
    assert rel_err(y, G(pre + "/train_out")) <= 2e-6
    assert G(pre + "/names") == ["B", "O"]
    (y * G(pre + "/gout")).sum().backward()
    assert rel_err(x.grad, G(pre + "/train_gx")) <= 2e-5
    for i in range(L):
        assert rel_err(kw["conv_weights"][i].grad, G(f"{pre}/train_gconv_w{i}")) <= 2e-5
        if use_bn:
            assert rel_err(kw["bn_weights"][i].grad, G(f"{pre}/train_gbn_w{i}")) <= 2e-5
            assert rel_err(kw["bn_running_means"][i], G(f"{pre}/run_mean{i}")) <= 2e-6
            assert rel_err(kw["bn_running_vars"][i], G(f"{pre}/run_var{i}")) <= 2e-6
    assert rel_err(kw["fc_weight"].grad, G(pre + "/train_gfc_w")) <= 2e-5
    # eval mode with the updated running statistics
    x2 = G(pre + "/x").requires_grad_()
    kw2 = {k: ([t.detach() if t is not None else None for t in v] if isinstance(v, list) else
               (v.detach() if torch.is_tensor(v) else v)) for k, v in kw.items()}
    y2 = O.cin_layer(x2, training=False, **kw2)
    assert rel_err(y2, G(pre + "/eval_out")) <= 2e-6
    (y2 * G(pre + "/gout")).sum().backward()
    assert rel_err(x2.grad, G(pre + "/eval_gx")) <= 2e-5


def _mlp(G, pre):
    ws, bs, i = [], [], 0
    while G.has(f"{pre}_w{i}"):
        ws.append(G(f"{pre}_w{i}"))
        bs.append(G(f"{pre}_b{i}"))
        i += 1
    return ws, bs


@pytest.mark.parametrize("shape", MODEL_SHAPES)
def test_models(golden, shape):
    G = golden("models")
    t = "model/" + _tag(shape)
    off = O.field_offsets(G(t + "/field_sizes").tolist())
    idx, gout = G(t + "/idx"), G(t + "/gout")

    def lookups():
        ew = G(t + "/emb_w").requires_grad_()
        fw = G(t + "/feat_w").requires_grad_()
        return ew, fw, O.multi_indices_embedding(ew, idx, off), O.multi_indices_embedding(fw, idx, off)

    ew, fw, emb, feat = lookups()
    y = O.fm_model(feat, emb, G(t + "/fm_bias"))
    assert rel_err(y, G(t + "/fm_out")) <= 2e-6
    (y * gout).sum().backward()
    assert rel_err(ew.grad, G(t + "/fm_gemb")) <= 1e-5
    assert rel_err(fw.grad, G(t + "/fm_gfeat")) <= 1e-5

    ew, fw, emb, feat = lookups()
    ws, bs = _mlp(G, t + "/deepfm")
    ws[0].requires_grad_()
    y = O.deepfm_model(feat, emb, ws, bs)
    assert rel_err(y, G(t + "/deepfm_out")) <= 2e-6
    (y * gout).sum().backward()
    assert rel_err(ew.grad, G(t + "/deepfm_gemb")) <= 1e-5
    assert rel_err(fw.grad, G(t + "/deepfm_gfeat")) <= 1e-5
    assert rel_err(ws[0].grad, G(t + "/deepfm_gw0")) <= 1e-5

    ew, fw, emb, feat = lookups()
    ws, bs = _mlp(G, t + "/dcn")
    cW = [w.clone().requires_grad_() for w in G(t + "/dcn_cross_W")]
    y = O.dcn_model(emb, cW, list(G(t + "/dcn_cross_b")), ws, bs, G(t + "/dcn_fc_w"), G(t + "/dcn_fc_b"))
    assert rel_err(y, G(t + "/dcn_out")) <= 2e-6
    (y * gout).sum().backward()
    assert rel_err(ew.grad, G(t + "/dcn_gemb")) <= 1e-5
    assert rel_err(torch.stack([w.grad for w in cW]), G(t + "/dcn_gcross_W")) <= 1e-5

    ew, fw, emb, feat = lookups()
    ws, bs = _mlp(G, t + "/xdfm")
    cin_kw = dict(
        conv_weights=[G(t + f"/xdfm_conv_w{i}") for i in range(2)],
        conv_biases=[G(t + f"/xdfm_conv_b{i}") for i in range(2)],
        bn_weights=[G(t + f"/xdfm_bn_w{i}") for i in range(2)],
        bn_biases=[G(t + f"/xdfm_bn_b{i}") for i in range(2)],
        fc_weight=G(t + "/xdfm_fc_w"), fc_bias=G(t + "/xdfm_fc_b"), training=True)
    y = O.xdeepfm_model(feat, emb, cin_kw, ws, bs, G(t + "/xdfm_bias"))
    assert rel_err(y, G(t + "/xdfm_out")) <= 5e-6
    (y * gout).sum().backward()
    assert rel_err(ew.grad, G(t + "/xdfm_gemb")) <= 5e-5
    assert rel_err(fw.grad, G(t + "/xdfm_gfeat")) <= 1e-5


# ---- OuterProductNetwork / AFM / BilinearInteraction (tests/golden/make_golden_pairs.py) ----
from conftest import PAIR_SHAPES, pair_heavy_ok  # noqa: E402


@pytest.mark.parametrize("shape", PAIR_SHAPES)
def test_outer_product_afm_bilinear(golden, shape):
    G = golden("pairs")
    B, N, E = shape
    t = _tag(shape)
    x = G(f"x/{t}")
    for kt in ("mat", "vec", "num"):
        if kt == "mat" and not pair_heavy_ok(N, E):
            assert not G.has(f"opn_mat/{t}/out")
            continue
        xa = x.clone().requires_grad_()
        k = G(f"opn_{kt}/{t}/kernel").requires_grad_()
        y = O.outer_product_layer(xa, k, kt)
        assert rel_err(y, G(f"opn_{kt}/{t}/out")) <= 2e-6
        assert G(f"opn_{kt}/{t}/names") == ["B", "O"]
        (y * G(f"opn_{kt}/{t}/gout")).sum().backward()
        assert rel_err(xa.grad, G(f"opn_{kt}/{t}/gx")) <= 2e-6
        assert rel_err(k.grad, G(f"opn_{kt}/{t}/gkernel")) <= 2e-6
    # AFM
    xa = G(f"afm/{t}/x").requires_grad_()
    ps = [G(f"afm/{t}/{n}").requires_grad_() for n in ("W1", "b1", "W2", "b2")]
    y, attn = O.afm_layer(xa, *ps)
    assert rel_err(y, G(f"afm/{t}/out")) <= TOL * 2
    assert rel_err(attn, G(f"afm/{t}/attn")) <= TOL * 2
    assert G(f"afm/{t}/names") == ["B", "E"] and G(f"afm/{t}/attn_names") == ["None"] * 3
    ((y * G(f"afm/{t}/gout")).sum() + (attn * G(f"afm/{t}/gattn")).sum()).backward()
    assert rel_err(xa.grad, G(f"afm/{t}/gx")) <= 5e-6
    for p_, n in zip(ps, ("gW1", "gb1", "gW2", "gb2")):
        assert rel_err(p_.grad, G(f"afm/{t}/{n}")) <= 1e-5, n
    # Bilinear
    for bt in ("all", "each"):
        if bt == "each" and not pair_heavy_ok(N, E):
            continue
        xa = x.clone().requires_grad_()
        W = G(f"bil_{bt}/{t}/W").requires_grad_()
        b = G(f"bil_{bt}/{t}/b").requires_grad_()
        y = O.bilinear_layer(xa, W, b, bt)
        assert rel_err(y, G(f"bil_{bt}/{t}/out")) <= 2e-6
        assert G(f"bil_{bt}/{t}/names") == ["B", "N", "O"]
        (y * G(f"bil_{bt}/{t}/gout")).sum().backward()
        assert rel_err(xa.grad, G(f"bil_{bt}/{t}/gx")) <= 2e-6
        assert rel_err(W.grad, G(f"bil_{bt}/{t}/gW")) <= 2e-6
        assert rel_err(b.grad, G(f"bil_{bt}/{t}/gb")) <= 2e-6


def test_pair_layer_quirks_in_reference(golden):
    """bias=False cannot be constructed in the reference (int64 Parameter, bilinear_interaction.py:62),
    'interaction' is NotImplemented (:214), an unknown OPN kernel type is a ValueError (outer_product_network.py:66)."""
    G = golden("pairs")
    assert G("raises/bil_nobias") == ["RuntimeError"]
    assert G("raises/bil_interaction") == ["NotImplementedError"]
    assert G("raises/opn_badtype") == ["ValueError"]
    with pytest.raises(ValueError):
        O.outer_product_layer(torch.zeros(2, 3, 4), torch.zeros(1, 3, 4), "cube")
    with pytest.raises(ValueError):
        O.bilinear_layer(torch.zeros(2, 3, 4), torch.zeros(4, 4), torch.zeros(4), "interaction")


# ---- AFM at the reference's default configuration: training mode, dropout on the scores and on the output
# (tests/golden/make_golden_afm_drop.py records the masks the reference drew) ----
from conftest import AFM_DROP_SHAPES  # noqa: E402


@pytest.mark.parametrize("shape", AFM_DROP_SHAPES)
def test_afm_training_mode_dropout(golden, shape):
    G = golden("afm_drop")
    t = _tag(shape)
    p = float(G(f"{t}/p")[0])
    scale = 1.0 / (1.0 - p)
    xa = G(f"{t}/x").requires_grad_()
    ps = [G(f"{t}/{n}").requires_grad_() for n in ("W1", "b1", "W2", "b2")]
    keep = G(f"{t}/score_keep")
    assert 0 < int(keep.sum()) < keep.numel()                       # the fixture really drops some scores
    y0, attn = O.afm_layer(xa, *ps, score_keep=keep, keep_scale=scale)
    assert rel_err(y0, G(f"{t}/out_before_dropout")) <= TOL * 2
    assert rel_err(attn, G(f"{t}/attn")) <= TOL * 2
    y = y0 * G(f"{t}/out_keep").to(y0.dtype) * scale                # the output nn.Dropout given its mask
    assert rel_err(y, G(f"{t}/out")) <= TOL * 2
    ((y * G(f"{t}/gout")).sum() + (attn * G(f"{t}/gattn")).sum()).backward()
    assert rel_err(xa.grad, G(f"{t}/gx")) <= 5e-6
    for p_, n in zip(ps, ("gW1", "gb1", "gW2", "gb2")):
        assert rel_err(p_.grad, G(f"{t}/{n}")) <= 1e-5, n
