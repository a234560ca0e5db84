"""Developer probe: AFM parameter gradients, MFMA path vs generic path vs fp32 oracle on one shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O
from torecsys_amd import functional as F_
B, N, E, A = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (17, 6, 64, 96)
dev = torch.device("cuda:0"); dt = torch.bfloat16
g = torch.Generator().manual_seed(B + N + E + A)
x0 = (torch.randn(B, N, E, generator=g) * 0.7).to(dt)
ps = [(torch.randn(A, E, generator=g) / E ** 0.5).to(dt), (torch.randn(A, generator=g) * 0.1).to(dt),
      (torch.randn(1, A, generator=g) / A ** 0.5).to(dt), (torch.randn(1, generator=g) * 0.1).to(dt)]
go, ga = torch.randn(B, E, generator=g), torch.randn(B, N * (N - 1) // 2, generator=g)
x = x0.to(dev).requires_grad_(); pd = [p.to(dev).requires_grad_() for p in ps]
y, attn = F_.afm(x, *pd)
((y.float() * go.to(dev)).sum() + (attn.float() * ga.to(dev)).sum()).backward()
xr = x0.float().requires_grad_(); pr = [p.float().requires_grad_() for p in ps]
yr, ar = O.afm_layer(xr, *pr)
((yr * go).sum() + (ar.squeeze(-1) * ga).sum()).backward()
def rel(a, b): return float((a.float().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
print("path", "generic" if os.environ.get("TRS_AFM_GENERIC") else "mfma", "out", rel(y, yr), "attn", rel(attn, ar.squeeze(-1)), "gx", rel(x.grad, xr.grad))
for n_, a, b in zip(("gW1", "gb1", "gw2", "gb2"), pd, pr):
    print(n_, rel(a.grad, b.grad), float(b.grad.abs().max()))
