"""one forward + backward of the DCN per-field MLP stack at the bench size (for rocprofv3 counter passes: tools/pmc_mlp.sh,
tools/pmc_run.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torecsys_amd import functional as F_
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(1); dt = torch.bfloat16
widths = [64, 400, 400, 400, 64]; rows = 65536 * 39
Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dt).to(dev) for i, o in zip(widths[:-1], widths[1:])]
bs = [(0.1 * torch.randn(o, generator=g)).to(dt).to(dev) for o in widths[1:]]
x = torch.randn(rows, widths[0], generator=g).to(dt).to(dev)
gy = torch.randn(rows, widths[-1], generator=g).to(dt).to(dev)
for _ in range(3):
    y, hidden, masks = F_.fused_mlp_forward_raw(x, Ws, bs)
    F_.fused_mlp_backward_raw(gy, widths, Ws, masks)
torch.cuda.synchronize()
