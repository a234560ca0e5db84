"""Developer probe: which pieces of a step survive hipGraph capture (each case in its own process)."""
import subprocess, sys, os
CASES = ["eager_first_w0", "eager_first_nopf", "eager_first_w2", "eager_first_w2_sync", "emb_fwdbwd", "feat_fwdbwd", "fm_only", "mlp_only", "bce", "full_bf16", "full_fp32", "full_big"]
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
    sys.exit(0)
case = sys.argv[1]
if case == "eager_first_nopf":
    os.environ["TRS_PREFETCH_BUCKETS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
from torecsys_amd import models as M, functional as F_
from torecsys_amd.graph import GraphedStep
dev = torch.device("cuda:0")
dt = torch.float32 if case in ("full_fp32", "emb_fwd", "emb_fwdbwd", "feat_fwdbwd", "fm_only", "mlp_only", "bce") else torch.bfloat16
B, N, E = (65536, 39, 64) if case == "full_big" else (512, 7, 32)
sizes = [25641] * N if case == "full_big" else [50, 3, 1000, 17, 400, 9, 121]
emb = MultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=True)
feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
emb.set_schema(["c0"]); feat.set_schema(["c0"])
inputs = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev).to(dt)
model = M.DeepFactorizationMachineModel(E, N, [64, 32], fm_dropout_p=0.0).to(dev).to(dt)
params = list(inputs.parameters()) + list(model.parameters())
g = torch.Generator().manual_seed(0)
ix = torch.stack([torch.randint(0, s, (B,), generator=g) for s in sizes], 1).to(dev)
lab = (torch.rand(B, 1, generator=g) < 0.3).float().to(dev)
crit = torch.nn.BCEWithLogitsLoss()
def fn(ix, lab):
    if case == "emb_fwd":
        with torch.no_grad():
            return emb(ix).rename(None).sum()
    if case == "emb_fwdbwd":
        l = emb(ix).rename(None).float().sum(); l.backward(); return l
    if case == "feat_fwdbwd":
        l = feat(ix).rename(None).float().sum(); l.backward(); return l
    if case == "fm_only":
        m = M.FactorizationMachineModel(E, N, dropout_p=0.0).to(dev).to(dt) if False else None
        d = inputs({"c0": ix}); l = (d["emb_inputs"].rename(None).float().sum() + d["feat_inputs"].rename(None).float().sum()); l.backward(); return l
    if case == "mlp_only":
        x = torch.ones(B, 1, N * E, device=dev, dtype=dt); l = model.deep(x).rename(None).float().sum(); l.backward(); return l
    if case == "bce":
        l = crit(lab * 0.5, lab); return l
    l = crit(model(**inputs({"c0": ix})).float(), lab); l.backward(); return l
if case.startswith("eager_first"):
    dt = torch.bfloat16 if case.endswith("bf16") else torch.float32
    inputs.to(dt); model.to(dt)
    keep = []
    for _ in range(3):
        for p in params: p.grad = None
        l = fn(ix, lab)
        if not case.endswith("noclone"):
            keep.append((l.detach().clone(), emb.embedding.weight.grad.clone()))
if case == "eager_first_w2_sync":
    torch.cuda.synchronize(); F_.clear_caches(); import gc; gc.collect(); torch.cuda.empty_cache()
step = GraphedStep(fn, (ix, lab), params=params, warmup=0 if case == "eager_first_w0" else 2)
out = step(ix, lab); torch.cuda.synchronize()
print("ok", float(out))
