"""World-size-2 (and 3) CPU runs of the row-sharded lookup on the gloo backend: host logic of
torecsys_amd/dist.py (owner bucketing contract, all-to-all choreography, un-permute, reverse exchange,
shard gradient) with the device ops replaced by torch-CPU stand-ins defined HERE (tests may use the
oracle; the product default is the HIP ops and refuses CPU tensors)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class CpuOps:
    """torch-CPU statement of the HipOps contract (tests only)."""

    def bucket_by_owner(self, idx, offsets, rows_per_rank, world):
        g = (idx.long() + offsets.view(1, -1)).reshape(-1)
        owner = torch.div(g, rows_per_rank, rounding_mode="floor").clamp_(0, world - 1)
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=world).to(torch.int64)
        send_ids = (g - owner * rows_per_rank)[order].to(torch.int32)
        send_pos = order.to(torch.int32)
        inv = torch.empty_like(send_pos)
        inv[order] = torch.arange(order.numel(), dtype=torch.int32)
        return counts, send_ids, send_pos, inv

    def gather_local(self, weight, ids, n_valid=None, padded=False, out=None):
        res = weight.detach()[ids.long().clamp(0, weight.shape[0] - 1)]
        if n_valid is not None:
            res = res * ((ids >= 0) & (ids < n_valid)).unsqueeze(-1).to(res.dtype)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def unpermute_local(self, back, weight, n_valid, inv_pos, send_ids, self_lo, self_n, B, N, want_fm, out=None):
        """HipOps.unpermute_local: slots [self_lo, self_lo + self_n) read this rank's own shard, the others the received
        rows, which are stored without that segment"""
        from oracle import cpu_ref as O
        s = inv_pos.long()
        E = weight.shape[1]
        is_local = (s >= self_lo) & (s < self_lo + self_n)
        rows = torch.zeros(s.numel(), E, dtype=weight.dtype)
        ids = send_ids.long()[s.clamp(0, max(send_ids.numel() - 1, 0))] if send_ids.numel() else torch.zeros_like(s)
        ok = is_local & (ids >= 0) & (ids < n_valid)
        rows[ok] = weight.detach()[ids[ok]]
        rem = ~is_local
        bi = s - (s >= self_lo + self_n).long() * self_n
        if rem.any():
            rows[rem] = back[bi[rem]]
        block = rows.reshape(B, N, E)
        if not want_fm:
            return block, None, None
        return block, O.fm_layer(block.float()).to(block.dtype), block.float().sum(1)

    def unique_route(self, idx, offsets, rows_per_rank, world):
        g = (idx.long() + offsets.view(1, -1)).reshape(-1)
        uniq, inv = torch.unique(g, return_inverse=True)
        owner = torch.div(uniq, rows_per_rank, rounding_mode="floor").clamp_(0, world - 1)
        return (torch.bincount(owner, minlength=world).to(torch.int64), (uniq - owner * rows_per_rank).to(torch.int32),
                inv.to(torch.int32))

    def reduce_grad_unique(self, g_block, inv, rows, g_fm, fm_sum):
        U, E = rows.shape
        K = inv.numel()
        g = torch.zeros(K, E) if g_block is None else g_block.reshape(K, E).float().clone()
        if g_fm is not None:
            N = K // g_fm.shape[0]
            x = rows[inv.long()].float().reshape(-1, N, E)
            g = g + (g_fm.unsqueeze(1).float() * (fm_sum.unsqueeze(1) - x)).reshape(K, E)
        out = torch.zeros(U, E)
        out.index_add_(0, inv.long(), g)
        return out.to(rows.dtype)

    def shard_update(self, weight, ids, grad_rows, opt, dense_index, padded=False):
        if padded:                       # -1 = padding slot of a fixed-capacity exchange: updates nothing
            keep = ids >= 0
            ids, grad_rows = ids[keep], grad_rows[keep]
        with torch.no_grad():
            g = torch.zeros_like(weight, dtype=torch.float32)
            g.index_add_(0, ids.long(), grad_rows.float())
            touched = torch.zeros(weight.shape[0], dtype=torch.bool)
            touched[ids.long()] = True
            w = weight.data
            if opt.kind == 1:
                w[touched] -= (opt.lr * g[touched]).to(w.dtype)
            elif opt.kind == 2:
                st = opt.state_for(w)
                st[touched] += g[touched] ** 2
                w[touched] -= (opt.lr * g[touched] / (st[touched].sqrt() + opt.eps)).to(w.dtype)
            else:
                raise NotImplementedError

    def unpermute(self, rows, inv_pos, B, N, want_fm):
        from oracle import cpu_ref as O
        block = rows[inv_pos.long()].reshape(B, N, -1)
        if not want_fm:
            return block, None, None
        return block, O.fm_layer(block.float()).to(block.dtype), block.float().sum(1)

    def permute_grad(self, g_block, send_pos, g_fm, fm_sum, block, out=None):
        if g_fm is not None:
            dx = g_fm.unsqueeze(1).float() * (fm_sum.unsqueeze(1) - block.float())
            g_block = dx.to(block.dtype) if g_block is None else g_block + dx.to(block.dtype)
        E = g_block.shape[-1]
        res = g_block.reshape(-1, E)[send_pos.long().clamp_min(0)]
        res = res * (send_pos >= 0).unsqueeze(-1).to(res.dtype)          # padding slots: zero rows
        if out is not None:
            out.copy_(res)
            return out
        return res

    def prefetch_own_buckets(self, weight, idx, offsets_local):
        pass

    def own_grad_dense(self, weight, idx, offsets_local, g_block, g_fm, fm_sum):
        """HipOps.own_grad_dense: dense gradient of the lookups that land inside this rank's shard, straight from the
        block gradient with the FM term folded in (x = the table row itself)"""
        V, E = weight.shape
        B, N = idx.shape
        r = (idx.long() + offsets_local.view(1, -1)).reshape(-1)
        ok = (r >= 0) & (r < V)
        g = torch.zeros(B * N, E) if g_block is None else g_block.reshape(-1, E).float().clone()
        if g_fm is not None:
            x = torch.zeros(B * N, E)
            x[ok] = weight.detach()[r[ok]].float()
            g = g + (g_fm.unsqueeze(1).float() * (fm_sum.unsqueeze(1) - x.reshape(B, N, E))).reshape(-1, E)
        out = torch.zeros(V, E)
        out.index_add_(0, r[ok], g[ok])
        return out.to(weight.dtype)

    def accumulate_rows(self, gw, ids, rows, padded=False):
        if padded:
            keep = ids >= 0
            ids, rows = ids[keep], rows[keep]
        gw.index_add_(0, ids.long(), rows.to(gw.dtype))

    def shard_grad_dense(self, weight, ids, grad_rows, padded=False):
        if padded:
            keep = ids >= 0
            ids, grad_rows = ids[keep], grad_rows[keep]
        g = torch.zeros_like(weight)
        g.index_add_(0, ids.long(), grad_rows)
        return g


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fuse, sparse, ret, dedup=False, optimizer=None, capacity=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cpu_ref as O
        from torecsys_amd.dist import RowShardedMultiIndicesEmbedding, shard_ranges
        torch.manual_seed(0)
        fs = [7, 3, 11, 5, 9]
        ragged = capacity is None                # fixed-capacity slots need the same batch size on every rank
        N, E = len(fs), 8
        V = sum(fs)
        g = torch.Generator().manual_seed(99)
        W = torch.randn(V, E, generator=g)
        idx_all = [torch.cat([torch.randint(0, f, (13 + (r if ragged else 0), 1), generator=g) for f in fs], 1) for r in range(world)]
        if dedup:                        # plenty of duplicate lookups inside every local batch
            for t in idx_all:
                t[1::2] = t[0::2][: t[1::2].shape[0]]
        gb_all = [torch.randn(13 + (r if ragged else 0), N, E, generator=g) for r in range(world)]
        gf_all = [torch.randn(13 + (r if ragged else 0), E, generator=g) for r in range(world)]
        m = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse, ops=CpuOps(),
                                            dense_grad_max_rows=0 if sparse else 10 ** 9, dedup=dedup,
                                            capacity=None if capacity is None else abs(capacity))
        opt = None
        if optimizer is not None:
            from torecsys_amd.optim import FusedSparseAdagrad, FusedSparseSGD
            opt = FusedSparseSGD(0.1) if optimizer == "sgd" else FusedSparseAdagrad(0.1, eps=1e-10)
            m.set_fused_optimizer(opt)
        assert list(m.state_dict().keys()) == ["embedding.weight"]
        m.load_full_weight(W)
        per, ranges = shard_ranges(V, world)
        assert m.row_range == ranges[rank] and m.rows_per_rank == per
        assert torch.equal(m.full_weight(), W)
        if fuse:                         # exercise the routed-ahead path (input-pipeline hint) on half the cases
            m.prefetch_route(idx_all[rank])
        if capacity is not None and capacity < 0:
            # too few slots on purpose: the lookups that do not fit read as zero rows and the device-side flag is raised
            from torecsys_amd import functional as F_
            F_.index_errors_seen()
            out = m(idx_all[rank])
            assert F_.index_errors_seen(), "slot overflow must raise the index flag"
            ret[rank] = "ok"
            return
        out = m(idx_all[rank])
        assert out.names == ("B", "N", "E")
        off = O.field_offsets(fs)
        ref = O.multi_indices_embedding(W, idx_all[rank], off)
        assert torch.equal(out.rename(None), ref), "sharded lookup must be bit-exact"
        loss = (out.rename(None) * gb_all[rank]).sum()
        if fuse:
            fm, ver = out._trs_fused_fm
            assert torch.allclose(fm, O.fm_layer(ref), rtol=1e-5, atol=1e-5)
            loss = loss + (fm * gf_all[rank]).sum()
        loss.backward()
        gw = m.embedding.weight.grad
        if opt is not None:
            assert gw is None, "the fused optimizer leaves no gradient tensor"
        elif sparse:
            assert gw.is_sparse
            gw = gw.to_dense()
        # reference: gradient of the full table summed over every rank's batch
        Wr = W.clone().requires_grad_()
        tot = 0
        for r in range(world):
            e = O.multi_indices_embedding(Wr, idx_all[r], off)
            tot = tot + (e * gb_all[r]).sum()
            if fuse:
                tot = tot + (O.fm_layer(e) * gf_all[r]).sum()
        tot.backward()
        lo, hi = m.row_range
        if opt is None:
            assert torch.allclose(gw[: hi - lo], Wr.grad[lo:hi], rtol=1e-5, atol=1e-5)
        else:
            # the owner's rows after its in-backward step == a dense torch.optim step on the full-table gradient
            Wd = W.clone().requires_grad_()
            Wd.grad = Wr.grad.clone()
            ref_opt = (torch.optim.SGD([Wd], lr=0.1) if optimizer == "sgd"
                       else torch.optim.Adagrad([Wd], lr=0.1, eps=1e-10))
            ref_opt.step()
            assert torch.allclose(m.embedding.weight.data[: hi - lo], Wd.data[lo:hi], rtol=1e-5, atol=1e-5)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(world, *args):
    assert dist.is_gloo_available()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, *args[:2], ret, *args[2:]), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


@pytest.mark.parametrize("world,fuse,sparse", [(2, False, False), (2, True, False), (2, True, True), (3, False, True)])
def test_row_sharded_lookup_gloo(world, fuse, sparse):
    _run(world, fuse, sparse)


@pytest.mark.parametrize("world,fuse,sparse,capacity", [(2, True, False, None), (3, False, True, None), (2, True, False, 2.0)])
def test_row_sharded_lookup_through_the_buffers_gloo(world, fuse, sparse, capacity, monkeypatch):
    """TRS_SHARD_LOCAL_DIRECT=0: the round-5 arrangement -- this rank's own lookups travel through the gather / exchange
    buffers like everybody else's (the default since round 6 reads them straight from the shard: every other test here
    runs that way) -- must keep producing the same block and gradients"""
    monkeypatch.setenv("TRS_SHARD_LOCAL_DIRECT", "0")
    _run(world, fuse, sparse, False, None, capacity)


@pytest.mark.parametrize("world,fuse,capacity", [(2, True, None), (3, False, None), (2, True, 2.0)])
def test_row_sharded_lookup_own_rows_permuted_gloo(world, fuse, capacity, monkeypatch):
    """TRS_SHARD_OWN_DIRECT=0: dense shard gradient with this rank's own gradient rows permuted behind the received ones
    and reduced together (the path a fused optimizer or a sparse gradient always takes) instead of the default, the
    unsharded backward on the rank's own lookups + accumulation of what arrived"""
    monkeypatch.setenv("TRS_SHARD_OWN_DIRECT", "0")
    _run(world, fuse, False, False, None, capacity)


@pytest.mark.parametrize("world,fuse,sparse", [(2, False, False), (2, True, False), (3, True, True)])
def test_row_sharded_lookup_dedup_gloo(world, fuse, sparse):
    """every distinct row id travels once: the block stays bit-exact, duplicate lookups' gradients are summed before
    the reverse exchange"""
    _run(world, fuse, sparse, True)


@pytest.mark.parametrize("world,fuse,sparse,dedup,optimizer", [(2, False, False, False, "sgd"), (2, True, True, False, "adagrad"),
                                                               (3, True, True, True, "sgd"), (1, True, False, False, "adagrad")])
def test_row_sharded_fused_optimizer_gloo(world, fuse, sparse, dedup, optimizer):
    """set_fused_optimizer on the sharded module: the owner steps its rows inside the backward pass (no gradient
    tensor), equal to a dense torch.optim step on the full table"""
    _run(world, fuse, sparse, dedup, optimizer)


@pytest.mark.parametrize("world,fuse,sparse,optimizer", [(2, False, False, None), (2, True, False, None), (3, True, True, None),
                                                         (2, True, False, "adagrad"), (3, False, False, "sgd")])
def test_row_sharded_fixed_capacity_gloo(world, fuse, sparse, optimizer):
    """capacity=...: equal-split all-to-alls over padded slots, no split size read on the host; block bit-exact, gradients /
    owner-side optimizer steps equal the unsharded reference (padding slots carry zero rows and update nothing)"""
    _run(world, fuse, sparse, False, optimizer, 2.0)


def test_row_sharded_fixed_capacity_overflow_is_flagged():
    """13 x 5 = 65 lookups over 2 ranks at factor 1.0 = 64 slots per peer rounded up: a skewed split overflows them"""
    from torecsys_amd.dist import pad_slots
    counts = torch.tensor([100, 28])
    ids = torch.arange(128, dtype=torch.int32)
    pos = torch.arange(128, dtype=torch.int32)
    ids_pad, pos_pad, inv_pad, bad = pad_slots(counts, ids, pos, pos.clone(), 64, 2)
    assert bool(bad) and ids_pad.shape == (128,)
    assert torch.equal(ids_pad[:64], ids[:64]) and torch.equal(ids_pad[64:92], ids[100:128])
    assert bool((ids_pad[92:] == -1).all()) and bool((pos_pad[92:] == -1).all())
    assert bool((inv_pad[64:100] == 128).all())                     # the 36 lookups that did not fit -> the zero row
    assert torch.equal(inv_pad[100:], torch.arange(64, 92, dtype=torch.int32))
    counts = torch.tensor([60, 64])
    _, _, inv_pad, bad = pad_slots(counts, ids[:124], pos[:124], pos[:124].clone(), 64, 2)
    assert not bool(bad) and torch.equal(inv_pad[60:], torch.arange(64, 128, dtype=torch.int32))


def _pipeline_worker(rank, world, port, fuse, capacity, ret):
    """>= 4 training-shaped steps with the cross-step pipeline (prefetch_lookup of batch k+1 issued before the backward of
    batch k; overlap_grad_exchange) against the same steps without it: blocks, FM terms and shard gradients bit-equal."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torecsys_amd import dist as D
        from torecsys_amd.dist import RowShardedMultiIndicesEmbedding
        fs = [7, 3, 11, 5, 9]
        N, E, B, STEPS = len(fs), 8, 12, 5
        g = torch.Generator().manual_seed(17)
        W = torch.randn(sum(fs), E, generator=g)
        batches = [[torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1) for _ in range(world)]
                   for _ in range(STEPS)]
        gbs = [[torch.randn(B, N, E, generator=g) for _ in range(world)] for _ in range(STEPS)]
        gfs = [[torch.randn(B, E, generator=g) for _ in range(world)] for _ in range(STEPS)]

        def run(pipelined):
            D.clear_route_caches()
            D._lookup_cache.clear()
            m = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse, ops=CpuOps(), capacity=capacity,
                                                overlap_grad_exchange=pipelined)
            m.load_full_weight(W)
            outs = []
            if pipelined:
                m.prefetch_lookup(batches[0][rank])
            for k in range(STEPS):
                m.embedding.weight.grad = None
                out = m(batches[k][rank])
                loss = (out.rename(None) * gbs[k][rank]).sum()
                fm = None
                if fuse:
                    fm = out._trs_fused_fm[0]
                    loss = loss + (fm * gfs[k][rank]).sum()
                if pipelined and k + 1 < STEPS:
                    m.prefetch_lookup(batches[k + 1][rank])        # the next batch's exchange, ahead of this backward
                loss.backward()
                m.wait_grad()
                outs.append((out.rename(None).detach().clone(), None if fm is None else fm.detach().clone(),
                             m.embedding.weight.grad.detach().clone()))
            return outs

        D.lookup_stats.update(prefetched=0, inline=0)
        a = run(False)
        assert D.lookup_stats["prefetched"] == 0 and D.lookup_stats["inline"] == STEPS
        D.lookup_stats.update(prefetched=0, inline=0)
        b = run(True)
        assert D.lookup_stats["prefetched"] == STEPS and D.lookup_stats["inline"] == 0, D.lookup_stats
        for k, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x[0], y[0]), f"block differs at step {k}"
            assert (x[1] is None and y[1] is None) or torch.equal(x[1], y[1]), f"FM differs at step {k}"
            assert torch.equal(x[2], y[2]), f"shard gradient differs at step {k}"
        # a fused optimizer makes an early lookup stale: refused unless asked for
        from torecsys_amd.optim import FusedSparseSGD
        m = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse, ops=CpuOps(), capacity=capacity)
        m.load_full_weight(W)
        m.set_fused_optimizer(FusedSparseSGD(0.1))
        try:
            m.prefetch_lookup(batches[0][rank])
            raise AssertionError("prefetch_lookup must refuse a fused optimizer without stale_ok")
        except RuntimeError as e:
            assert "stale_ok" in str(e)
        # stale_ok: step k+1 sees the rows as they were BEFORE step k's update (one update behind), nothing else changes
        m.prefetch_lookup(batches[0][rank], stale_ok=True)
        out0 = m(batches[0][rank])
        m.prefetch_lookup(batches[1][rank], stale_ok=True)
        (out0.rename(None) * gbs[0][rank]).sum().backward()          # updates the shards
        out1 = m(batches[1][rank])
        from oracle import cpu_ref as O
        ref1 = O.multi_indices_embedding(W, batches[1][rank], O.field_offsets(fs))
        assert torch.equal(out1.rename(None), ref1), "the early lookup reads the rows of before the update"
        # an ORDINARY in-place update of the shard (torch.optim's step) between the hint and the forward: the early rows
        # are dropped and the forward looks up again (round-4 advisor finding: it silently used the old rows)
        m2 = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=fuse, ops=CpuOps(), capacity=capacity)
        m2.load_full_weight(W)
        m2.prefetch_lookup(batches[2][rank])
        with torch.no_grad():
            m2.embedding.weight.mul_(2.0)
        D.lookup_stats.update(prefetched=0, inline=0, stale_dropped=0)
        out2 = m2(batches[2][rank])
        assert D.lookup_stats["stale_dropped"] == 1 and D.lookup_stats["inline"] == 1, D.lookup_stats
        assert torch.equal(out2.rename(None), O.multi_indices_embedding(2.0 * W, batches[2][rank], O.field_offsets(fs)))
        ret[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fuse,capacity", [(2, True, None), (3, False, None), (2, True, 2.0), (1, True, None)])
def test_row_sharded_pipelined_step_is_bit_equal(world, fuse, capacity):
    """dist.prefetch_lookup + overlap_grad_exchange (the cross-step pipeline of DESIGN.md section 6) change WHEN the
    exchanges are issued, never what they compute"""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, fuse, capacity, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def test_default_ops_refuse_cpu():
    from torecsys_amd.dist import HipOps
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipOps().bucket_by_owner(torch.zeros(2, 2, dtype=torch.long), torch.zeros(2, dtype=torch.long), 4, 2)


def _bucket_worker(rank, world, port, dtype, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torecsys_amd.dist import DenseGradBucket
        torch.manual_seed(0)
        shapes = [(400, 2496), (400,), (400, 400), (1, 400), (1,)]
        params = [torch.nn.Parameter(torch.zeros(*s, dtype=dtype)) for s in shapes]
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
        bucket = DenseGradBucket(params + [frozen])
        assert bucket.flat.dtype == dtype and bucket.flat.numel() == sum(p.numel() for p in params)
        assert bucket.bytes_per_step == (0 if world == 1 else bucket.flat.numel() * bucket.flat.element_size())
        for step in range(3):                                   # the same bucket, step after step
            g = torch.Generator().manual_seed(100 * step)
            all_grads = [[torch.randn(*s, generator=g).to(dtype) for s in shapes] for _ in range(world)]
            for p, gr in zip(params, all_grads[rank]):
                p.grad = gr.clone()
            bucket.reduce()
            bucket.wait()
            for i, p in enumerate(params):
                want = sum(all_grads[r][i].float() for r in range(world)) / world
                err = float((p.grad.float() - want).abs().max())
                tol = 1e-6 if dtype == torch.float32 else 2e-2 * float(want.abs().max()) + 1e-3
                assert err <= tol, (step, i, err)
        params[1].grad = None
        if world > 1:
            try:
                bucket.reduce()
                raise AssertionError("a missing gradient must be refused")
            except RuntimeError as e:
                assert "no gradient" in str(e)
        ret[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype", [(2, torch.float32), (2, torch.bfloat16), (3, torch.bfloat16), (1, torch.bfloat16)])
def test_dense_grad_bucket_averages_over_ranks(world, dtype):
    """dist.DenseGradBucket: one flat persistent bucket, the parameters' own dtype on the wire (bf16 for a bf16 model),
    gradients averaged in place; world 1 is a no-op"""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_bucket_worker, args=(world, port, dtype, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)
