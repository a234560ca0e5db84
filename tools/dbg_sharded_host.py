"""Developer probe: host time per phase of the row-sharded DeepFM step (world size 1), no profiler."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import bench
sys.argv = ["bench.py"]
a = bench.parse()
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from torecsys_amd import models as M
from torecsys_amd.inputs import Inputs
from torecsys_amd.dist import RowShardedMultiIndicesEmbedding
B, N, E = 65536, 39, 64
sizes = bench.field_sizes(1_000_000, N)
g = torch.Generator().manual_seed(0)
ring = [bench.synth_indices(B, sizes, g, False).to(dev) for _ in range(4)]
lab = (torch.rand(B, 1, generator=g) < 0.25).float().to(dev)
dt = torch.bfloat16
emb = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=True, dtype=dt, device=dev)
feat = RowShardedMultiIndicesEmbedding(embed_size=1, field_sizes=sizes, dtype=dt, device=dev)
emb.set_schema(["c0"]); feat.set_schema(["c0"])
inputs = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev).to(dt)
model = M.DeepFactorizationMachineModel(E, N, [400, 400, 400], fm_dropout_p=0.0).to(dev).to(dt)
crit = torch.nn.BCEWithLogitsLoss()
params = list(inputs.parameters()) + list(model.parameters())
T = {"inputs": 0.0, "model": 0.0, "loss": 0.0, "prefetch": 0.0, "backward": 0.0}
def step(k, rec):
    for p in params: p.grad = None
    t0 = time.perf_counter(); d = inputs({"c0": ring[k % 4]})
    t1 = time.perf_counter(); out = model(**d)
    t2 = time.perf_counter(); loss = crit(out.float(), lab)
    t3 = time.perf_counter(); emb.prefetch_route(ring[(k + 2) % 4])
    t4 = time.perf_counter(); loss.backward()
    t5 = time.perf_counter()
    if rec:
        for n_, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[n_] += v
if "--time-kernel" in sys.argv:
    from torecsys_amd import _abi
    _abi.time_kernel("trs_embed_fm", True)
for k in range(5): step(k, False)
import cProfile, pstats
pr_ = cProfile.Profile() if "--cprofile" in sys.argv else None
torch.cuda.synchronize(); t = time.perf_counter()
if pr_: pr_.enable()
for k in range(20): step(k, True)
if pr_: pr_.disable()
torch.cuda.synchronize(); el = time.perf_counter() - t
if pr_:
    pstats.Stats(pr_).sort_stats("tottime").print_stats(22)
print("ms/step", round(el / 20 * 1e3, 3), {k_: round(v / 20 * 1e3, 3) for k_, v in T.items()})
dist.destroy_process_group()
