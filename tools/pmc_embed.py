"""Run the fused lookup+FM forward kernel a few times at the BASELINE shape (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torecsys_amd import functional as F_
dev = torch.device("cuda:0")
B, N, E, V = 65536, 39, 64, 1_000_000
zipf = "--zipf" in sys.argv
g = torch.Generator().manual_seed(1234)
per = V // N
fs = [per] * (N - 1) + [V - per * (N - 1)]
off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.tensor(fs), 0)[:-1]]).to(dev)
if zipf:
    idx = torch.cat([(torch.pow(float(f), torch.rand(B, 1, generator=g, dtype=torch.float64)) - 1).clamp_(0, f - 1).long() for f in fs], 1).to(dev)
else:
    idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev)
w = torch.randn(V, E, generator=g).bfloat16().to(dev)
for _ in range(6):
    F_._EmbedFM.apply(w, idx, off, None, True)
torch.cuda.synchronize()
