"""Index staging: host-side packing logic (CPU) and the device-side paths (GPU)."""
import numpy as np
import pytest
import torch


def test_pack_host_layouts():
    from torecsys_amd.staging import pack_host
    B, N = 6, 4
    rng = np.random.default_rng(0)
    ref = rng.integers(0, 1000, (B, N))
    out = np.zeros((B, N), dtype=np.int32)
    assert np.array_equal(pack_host(ref, out), ref)                                   # (B,N) array
    assert np.array_equal(pack_host(torch.from_numpy(ref), out), ref)                 # CPU tensor
    assert np.array_equal(pack_host([ref[:, j].tolist() for j in range(N)], out), ref)   # list of per-field lists
    assert np.array_equal(pack_host([ref[:, :1], ref[:, 1], ref[:, 2:]], out), ref)   # mixed (B,), (B,k)
    d = {f"c{j}": torch.from_numpy(ref[:, j].copy()) for j in range(N)}
    assert np.array_equal(pack_host(d, out), ref)
    assert np.array_equal(pack_host(d, out, names=["c3", "c2", "c1", "c0"]), ref[:, ::-1])
    assert np.array_equal(pack_host([ref[:, j].astype(np.float32) for j in range(N)], out), ref)   # integral floats
    with pytest.raises(OverflowError):
        pack_host([np.array([2 ** 31] * B)] + [ref[:, j] for j in range(1, N)], out)
    with pytest.raises(ValueError):
        pack_host([ref[:, 0]], out)
    with pytest.raises(ValueError):
        pack_host([ref[:3, j] for j in range(N)], out)
    with pytest.raises(TypeError):
        pack_host([ref[:, j] + 0.5 for j in range(N)], out)
    out64 = np.zeros((B, N), dtype=np.int64)
    big = ref.astype(np.int64) + 2 ** 40
    assert np.array_equal(pack_host(big, out64), big)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.int64, torch.int32])
def test_pack_columns_equals_cat(dt):
    from torecsys_amd import functional as F_
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for B in (1, 63, 64, 1000, 4097):
        cols = [torch.randint(0, 10 ** 6, (B,), generator=g).to(dt).to(dev) for _ in range(5)]
        cols.insert(2, torch.randint(0, 10 ** 6, (B, 3), generator=g).to(dt).to(dev))
        ref = torch.cat([c.unsqueeze(-1) if c.dim() == 1 else c for c in cols], dim=1)
        out = F_.pack_columns(cols)
        assert out.dtype == dt and torch.equal(out, ref)
        assert torch.equal(F_.pack_columns(cols, out_dtype=torch.int32), ref.int())
    wide = [torch.arange(10, device=dev, dtype=dt) + j for j in range(39)]
    assert torch.equal(F_.pack_columns(wide), torch.stack(wide, 1))
    assert not F_.pack_columns_supported([wide[0]])
    assert not F_.pack_columns_supported([wide[0], wide[1].float()])
    assert not F_.pack_columns_supported([wide[0].cpu(), wide[1].cpu()])
    with pytest.raises(ValueError):
        F_.pack_columns([wide[0], wide[1][:5]])


@pytest.mark.gpu
def test_inputs_router_packs_once_and_shares_the_index_tensor():
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    dev = torch.device("cuda:0")
    sizes = [11, 7, 300, 5]
    N = len(sizes)
    emb = MultiIndicesEmbedding(embed_size=8, field_sizes=sizes)
    feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
    names = [f"f{i}" for i in range(N)]
    emb.set_schema(names); feat.set_schema(names)
    inputs = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev)
    g = torch.Generator().manual_seed(1)
    idx = torch.stack([torch.randint(0, s, (50,), generator=g) for s in sizes], 1).to(dev)
    seen = []
    h1 = emb.register_forward_pre_hook(lambda m, a: seen.append(a[0]))
    h2 = feat.register_forward_pre_hook(lambda m, a: seen.append(a[0]))
    out = inputs({n: idx[:, i].contiguous() for i, n in enumerate(names)})
    h1.remove(); h2.remove()
    assert seen[0] is seen[1] and torch.equal(seen[0], idx)
    off = torch.tensor([0, 11, 18, 318], device=dev)
    assert torch.equal(out["emb_inputs"].rename(None), emb.embedding.weight[idx + off])
    assert torch.equal(out["feat_inputs"].rename(None), feat.embedding.weight[idx + off])


@pytest.mark.gpu
def test_index_stager_roundtrip_and_ring():
    from torecsys_amd.staging import IndexStager
    dev = torch.device("cuda:0")
    B, N = 2048, 39
    st = IndexStager(B, N, dev, depth=2)
    rng = np.random.default_rng(3)
    batches = [rng.integers(0, 2 ** 31 - 1, (B, N)) for _ in range(5)]
    staged = [st.stage([b[:, j] for j in range(N)]) for b in batches]        # ring of 2 reused 5 times
    for s, b in zip(staged, batches):
        t = s.wait()
        assert t.dtype == torch.int32 and t.device.type == "cuda"
        assert np.array_equal(t.cpu().numpy(), b.astype(np.int32))
    st64 = IndexStager(B, N, dev, dtype=torch.int64)
    big = batches[0].astype(np.int64) + 2 ** 33
    assert np.array_equal(st64.stage(big).wait().cpu().numpy(), big)
    with pytest.raises(OverflowError):
        st.stage(big)
    with pytest.raises(RuntimeError):
        IndexStager(B, N, "cpu")
