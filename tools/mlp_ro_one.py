import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torecsys_amd import functional as F_
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(1); dt = torch.bfloat16
widths = [64, 400, 400, 400, 64]; rows = 65536 * 39
Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dt).to(dev) for i, o in zip(widths[:-1], widths[1:])]
bs = [(0.1 * torch.randn(o, generator=g)).to(dt).to(dev) for o in widths[1:]]
x = torch.randn(rows, widths[0], generator=g).to(dt).to(dev)
for _ in range(3):
    F_.fused_mlp_forward_raw(x, Ws, bs)
torch.cuda.synchronize()
