"""hipGraph capture of a whole step (torecsys_amd.graph.GraphedStep): replayed steps must reproduce eager steps
(forward bit for bit; gradients up to the summation order inside a row bucket, which the bucket build does not fix),
and the device-timestamp marks that time a kernel inside a graph must agree with HIP events."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _deepfm(dev, dt, N, E, sizes):
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    from harness import ctr_models as M
    torch.manual_seed(3)
    emb = MultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=True)
    feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
    emb.set_schema(["c0"]); feat.set_schema(["c0"])
    inputs = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev).to(dt)
    model = M.DeepFactorizationMachineModel(E, N, [64, 32], fm_dropout_p=0.0).to(dev).to(dt)
    return inputs, model, emb, feat


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_graphed_step_matches_eager(dev, dt):
    from torecsys_amd.graph import GraphedStep
    B, N, E = 512, 7, 32
    sizes = [50, 3, 1000, 17, 400, 9, 121]
    inputs, model, emb, feat = _deepfm(dev, dt, N, E, sizes)
    params = list(inputs.parameters()) + list(model.parameters())
    g = torch.Generator().manual_seed(0)
    batches = [(torch.stack([torch.randint(0, s, (B,), generator=g) for s in sizes], 1).to(dev),
                (torch.rand(B, 1, generator=g) < 0.3).float().to(dev)) for _ in range(3)]
    crit = torch.nn.BCEWithLogitsLoss()

    def fn(ix, lab):
        loss = crit(model(**inputs({"c0": ix})).float(), lab)
        loss.backward()
        return loss

    eager = []
    for ix, lab in batches:
        for p in params:
            p.grad = None
        loss = fn(ix, lab)
        eager.append((loss.detach().clone(), emb.embedding.weight.grad.clone(), feat.embedding.weight.grad.clone(),
                      model.deep.model.Linear_0.weight.grad.clone()))
    del loss        # a live autograd graph of an eager step pins the AccumulateGrad nodes to the eager stream
    step = GraphedStep(fn, batches[0], params=params, warmup=2)
    for (ix, lab), (l0, ge, gf, gw) in zip(batches, eager):
        loss = step(ix, lab)
        torch.cuda.synchronize()
        assert torch.equal(loss.detach(), l0)
        tol = 1e-5 if dt == torch.float32 else 1e-2      # the path's fp32 / bf16 tolerances (summation order)
        assert rel_err(emb.embedding.weight.grad.float().cpu(), ge.float().cpu()) <= tol
        assert rel_err(feat.embedding.weight.grad.float().cpu(), gf.float().cpu()) <= tol
        assert rel_err(model.deep.model.Linear_0.weight.grad.float().cpu(), gw.float().cpu()) <= tol
    with pytest.raises(ValueError):
        step(batches[0][0][:10], batches[0][1][:10])


def test_graphed_step_input_sets_replay_without_copies(dev):
    """GraphedStep(input_sets=K): one captured graph per set of static input buffers in one memory pool.  ``load(k, batch)``
    once, then ``replay(k)`` in any order and repeatedly: loss and gradients of the batch that set holds equal the eager
    step's, and ``p.grad`` is the tensor the replay has just refreshed."""
    from torecsys_amd.graph import GraphedStep
    dt = torch.bfloat16
    B, N, E = 512, 7, 32
    sizes = [50, 3, 1000, 17, 400, 9, 121]
    inputs, model, emb, feat = _deepfm(dev, dt, N, E, sizes)
    params = list(inputs.parameters()) + list(model.parameters())
    g = torch.Generator().manual_seed(1)
    batches = [(torch.stack([torch.randint(0, s, (B,), generator=g) for s in sizes], 1).to(dev),
                (torch.rand(B, 1, generator=g) < 0.3).float().to(dev)) for _ in range(3)]
    crit = torch.nn.BCEWithLogitsLoss()

    def fn(ix, lab):
        loss = crit(model(**inputs({"c0": ix})).float(), lab)
        loss.backward()
        return loss

    eager = []
    for ix, lab in batches:
        for p in params:
            p.grad = None
        loss = fn(ix, lab)
        eager.append((loss.detach().clone(), emb.embedding.weight.grad.clone(), model.deep.model.Linear_0.weight.grad.clone()))
    del loss
    step = GraphedStep(fn, batches[0], params=params, warmup=1, input_sets=3)
    assert step.input_sets == 3
    for k, (ix, lab) in enumerate(batches):
        step.load(k, ix, lab)
    for k in (2, 0, 1, 1, 2, 0):
        loss = step.replay(k)
        torch.cuda.synchronize()
        l0, ge, gw = eager[k]
        assert torch.equal(loss.detach(), l0), k
        assert rel_err(emb.embedding.weight.grad.float().cpu(), ge.float().cpu()) <= 1e-2, k
        assert rel_err(model.deep.model.Linear_0.weight.grad.float().cpu(), gw.float().cpu()) <= 1e-2, k
    # the copying form still works (set 0)
    loss = step(*batches[1])
    torch.cuda.synchronize()
    assert torch.equal(loss.detach(), eager[1][0])
    step.release_outputs()


def test_timestamp_marks_agree_with_events(dev):
    """_abi.time_kernel: HIP events in eager mode, trs_mark_timestamp pairs inside a capture."""
    from torecsys_amd import _abi
    from torecsys_amd import functional as F_
    B, N, E, V = 32768, 39, 64, 200_000
    w = torch.randn(V, E, device=dev, dtype=torch.bfloat16)
    idx = torch.randint(0, V // N, (B, N), device=dev)
    off = (torch.arange(N, device=dev) * (V // N))
    assert _abi.size_query("trs_wall_clock_khz") > 0

    def launch():
        return F_.embed_fm(w, idx, off)

    for _ in range(3):
        launch()
    _abi.time_kernel("trs_embed_fm", True)
    try:
        for _ in range(5):
            launch()
        ev = _abi.kernel_times_ms("trs_embed_fm")
        assert len(ev) == 5
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = launch()
        for _ in range(5):
            gr.replay()
        mk = _abi.kernel_times_ms("trs_embed_fm")
        assert len(mk) == 5
    finally:
        _abi.time_kernel("trs_embed_fm", False)
    a, b = sorted(ev)[2], sorted(mk)[2]
    assert 0.4 * a <= b <= 2.0 * a + 0.02, (ev, mk)
    assert out[0].shape == (B, N, E)


def test_graphed_dcn_mlp_stack_row_owner_replays_match_eager(dev):
    """The per-field MLP of DeepAndCrossNetwork at >= 131 072 rows (the row-owner kernels: their backward accumulates the
    bias gradients' column-sum partials with atomics on a buffer zeroed IN the captured work -- by a kernel, never a
    memset node): 5 replays of one captured forward + backward give the eager step's bias gradients bit for bit."""
    from torecsys_amd import functional as F_
    from torecsys_amd.graph import GraphedStep
    from torecsys_amd.layers import DNNLayer
    torch.manual_seed(9)
    B, N, E = 3400, 39, 64                     # 132 600 rows: AUTO picks the row-owner family
    lay = DNNLayer(inputs_size=E, output_size=E, layer_sizes=[400, 400, 400]).to(dev).bfloat16()
    assert F_.mlp_fused_family([E, 400, 400, 400, E], B * N) == F_.MLP_FAMILY_ROW_OWNER
    params = list(lay.parameters())
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(B, N, E, generator=g).bfloat16().to(dev) for _ in range(2)]
    gy = torch.randn(B, N, E, generator=g).bfloat16().to(dev)

    def fn(x):
        y = lay(x).rename(None)
        y.backward(gy)
        return y

    eager = []
    for x in xs:
        for p in params:
            p.grad = None
        y = fn(x)
        eager.append([y.detach().clone()] + [p.grad.clone() for p in params])
    del y
    step = GraphedStep(fn, (xs[0],), params=params, warmup=1)
    for rep in range(5):
        x, want = xs[rep % 2], eager[rep % 2]
        y = step(x)
        torch.cuda.synchronize()
        assert torch.equal(y.detach(), want[0]), rep
        for (name, p), w in zip(lay.named_parameters(), want[1:]):
            if name.endswith("bias"):
                assert torch.equal(p.grad, w), (rep, name)
            else:
                assert rel_err(p.grad.float().cpu(), w.float().cpu()) <= 1e-2, (rep, name)
