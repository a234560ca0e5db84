"""Worker of tests/test_gpu_dist.py::test_whole_sharded_step_with_rccl_inside_one_hipgraph (run as a script, in its own
process, under TRS_SHARD_FORCE_COLLECTIVES=1): captures the row-sharded lookup + FM + backward of a one-rank RCCL group --
all-to-alls included -- into one hipGraph, replays three batches and compares with the eager run.  Prints
"RCCL-IN-GRAPH OK" on success; any assertion fails the process."""
import os
import socket

import torch
import torch.distributed as dist


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from torecsys_amd import dist as D
    from torecsys_amd.dist import RowShardedMultiIndicesEmbedding
    from torecsys_amd.graph import GraphedStep
    from torecsys_amd.layers import FMLayer
    assert D.FORCE_COLLECTIVES, "run under TRS_SHARD_FORCE_COLLECTIVES=1"
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    B, N, E = 4096, 39, 64
    fs = [300 + 11 * i for i in range(N)]
    m = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=True, dtype=torch.bfloat16, device=dev,
                                        local_direct=False)
    batches = [torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev) for _ in range(3)]
    gb = (torch.randn(B, N, E, generator=g) * 0.1).bfloat16().to(dev)
    held = {}

    def fn(ix):
        out = m(ix)
        y = FMLayer()(out)
        loss = (out.rename(None).float() * gb.float()).sum() + (y.rename(None).float() ** 2).sum()
        loss.backward()
        held["out"], held["fm"] = out.rename(None).detach(), y.rename(None).detach()
        return loss

    eager = []
    for ix in batches:
        m.embedding.weight.grad = None
        fn(ix)
        torch.cuda.synchronize()
        eager.append((held["out"].clone(), held["fm"].clone(), m.embedding.weight.grad.clone()))
    m.embedding.weight.grad = None
    held.clear()
    D.clear_route_caches()
    step = GraphedStep(fn, (batches[0],), params=[m.embedding.weight], warmup=1)
    for k, ix in enumerate(batches):
        step(ix)
        torch.cuda.synchronize()
        assert torch.equal(held["out"], eager[k][0]), k
        assert torch.equal(held["fm"], eager[k][1]), k
        a, b = m.embedding.weight.grad.float(), eager[k][2].float()
        err = float((a - b).abs().max() / b.abs().max())
        assert err <= 4e-3, (k, err)      # (bucket order is not fixed: one bf16 ulp)
    print("RCCL-IN-GRAPH OK", flush=True)
    os._exit(0)      # (skip the process group's teardown: nothing after this line is under test)


if __name__ == "__main__":
    main()
