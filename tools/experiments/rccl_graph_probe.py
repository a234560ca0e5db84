"""Developer probe (round 6): which ingredient of bench.py's whole-step capture stalls when the one-rank group is forced to
issue its all-to-alls through RCCL (TRS_SHARD_FORCE_COLLECTIVES=1)?  usage: python rccl_graph_probe.py <variant>
variants: one (one table, as tests/rccl_graph_worker.py), two (E = 64 table + E = 1 table sharing the indices / the route),
two_big (the same at B = 65 536), inputs (both tables behind the Inputs router)."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main(variant):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from torecsys_amd import dist as D
    from torecsys_amd.dist import RowShardedMultiIndicesEmbedding
    from torecsys_amd.graph import GraphedStep
    from torecsys_amd.inputs import Inputs
    from torecsys_amd.layers import FMLayer
    assert D.FORCE_COLLECTIVES
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    B = 65536 if variant == "two_big" else 4096
    N, E = 39, 64
    fs = [300 + 11 * i for i in range(N)]
    m = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=fs, fuse_fm=True, dtype=torch.bfloat16, device=dev,
                                        local_direct=False)
    f1 = RowShardedMultiIndicesEmbedding(embed_size=1, field_sizes=fs, dtype=torch.bfloat16, device=dev, local_direct=False)
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev)
    inp = None
    if variant == "inputs":
        m.set_schema(["c0"]); f1.set_schema(["c0"])
        inp = Inputs(schema={"emb_inputs": m, "feat_inputs": f1})

    def fn(ix):
        if inp is not None:
            d = inp({"c0": ix})
            out, first = d["emb_inputs"], d["feat_inputs"]
        else:
            out = m(ix)
            first = f1(ix) if variant != "one" else None
        y = FMLayer()(out)
        loss = (out.rename(None).float() ** 2).sum() + (y.rename(None).float() ** 2).sum()
        if first is not None:
            loss = loss + first.rename(None).float().sum()
        loss.backward()
        return loss

    params = [m.embedding.weight] + ([f1.embedding.weight] if variant != "one" else [])
    for _ in range(2):
        for p in params:
            p.grad = None
        fn(idx)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    print(variant, "eager ok", flush=True)
    D.clear_route_caches()
    step = GraphedStep(fn, (idx,), params=params, warmup=1)
    print(variant, "captured", flush=True)
    for _ in range(3):
        step(idx)
    torch.cuda.synchronize()
    print(variant, "REPLAYED OK", flush=True)
    os._exit(0)


if __name__ == "__main__":
    main(sys.argv[1])
