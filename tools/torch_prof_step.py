"""Developer tool: torch.profiler table of one DeepFM bench step (which ATen ops own the non-GEMM kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (bench sets the TunableOp env before torch loads; pass --no-tunableop to skip)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "3", "--no-cpu-baseline"] + sys.argv[1:]
from torch.profiler import profile, ProfilerActivity
import torch.nn as nn
a = bench.parse()
dev = torch.device("cuda:0")
from torecsys_amd import models as M
from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
B, N, E = a.batch, a.fields, a.embed
sizes = bench.field_sizes(1_000_000, N)
g = torch.Generator().manual_seed(0)
idx = bench.synth_indices(B, sizes, g, False).to(dev)
lab = (torch.rand(B, 1, generator=g) < 0.25).float().to(dev)
emb = MultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=True); feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
emb.set_schema(["c0"]); feat.set_schema(["c0"])
inputs = Inputs({"emb_inputs": emb, "feat_inputs": feat}).to(dev).bfloat16()
model = M.DeepFactorizationMachineModel(E, N, [400, 400, 400], fm_dropout_p=0.0).to(dev).bfloat16()
crit = nn.BCEWithLogitsLoss()
def step():
    for p in list(inputs.parameters()) + list(model.parameters()): p.grad = None
    out = model(**inputs({"c0": idx})); crit(out.float(), lab).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
