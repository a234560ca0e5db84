import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O
from torecsys_amd.layers import CompressInteractionNetworkLayer
from torecsys_amd import functional as F_
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
B, N, E, sizes = 70, 39, 64, [64, 64]
torch.manual_seed(B + N + E)
lay = CompressInteractionNetworkLayer(embed_size=E, num_fields=N, output_size=2, layer_sizes=sizes).to(dev).bfloat16().train()
g = torch.Generator().manual_seed(5)
x = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16()
go = torch.randn(B, 2, generator=g)
kw = dict(conv_weights=[s.Conv1d.weight.detach().float().cpu().requires_grad_() for s in lay.model],
          conv_biases=[s.Conv1d.bias.detach().float().cpu() for s in lay.model],
          bn_weights=[s.Batchnorm.weight.detach().float().cpu() for s in lay.model],
          bn_biases=[s.Batchnorm.bias.detach().float().cpu() for s in lay.model],
          fc_weight=lay.fc.weight.detach().float().cpu(), fc_bias=lay.fc.bias.detach().float().cpu(), training=True)
xr = x.float().requires_grad_()
yr = O.cin_layer(xr, **kw); (yr * go).sum().backward()
for mode in ("cl", "generic"):
    lay.zero_grad()
    xd = x.to(dev).requires_grad_()
    if mode == "generic":
        old = F_.cin_cl_supported
        F_.cin_cl_supported = lambda *a, **k: False
    y = lay(xd)
    (y.rename(None).float() * go.to(dev)).sum().backward()
    if mode == "generic":
        F_.cin_cl_supported = old
    print(mode, "out", rel(y.rename(None).float().cpu(), yr.detach()), "gx", rel(xd.grad.float().cpu(), xr.grad),
          "gW0", rel(lay.model[0].Conv1d.weight.grad.float().cpu(), kw["conv_weights"][0].grad),
          "gW1", rel(lay.model[1].Conv1d.weight.grad.float().cpu(), kw["conv_weights"][1].grad))
# single contraction check CL vs channels-first generic, fwd+bwd
H, C = 32, 64
x0 = (0.5 * torch.randn(B, N, E, generator=g)).bfloat16().to(dev)
xk = (0.5 * torch.randn(B, H, E, generator=g)).bfloat16().to(dev)
W = (torch.randn(C, N * H, generator=g) / (N * H) ** 0.5).bfloat16().to(dev)
bias = (0.1 * torch.randn(C, generator=g)).bfloat16().to(dev)
gy = torch.randn(B, C, E, generator=g).bfloat16().to(dev)
a = [t.clone().requires_grad_() for t in (x0, xk, W, bias)]
y1 = F_.cin_contract(*a); (y1.float() * gy.float()).sum().backward()
ld0 = 64
x0T = torch.zeros(B, E, ld0, dtype=torch.bfloat16, device=dev); x0T[:, :, :N] = x0.transpose(1, 2)
x0T.requires_grad_()
xkT = xk.transpose(1, 2).contiguous().requires_grad_()
Wb, bb = W.clone().requires_grad_(), bias.clone().requires_grad_()
y2 = F_.cin_contract_cl(x0T, xkT, Wb, bb, N, H); (y2.float() * gy.transpose(1, 2).float()).sum().backward()
print("contract fwd", rel(y2.transpose(1, 2).float(), y1.float()), "gx0", rel(x0T.grad[:, :, :N].transpose(1, 2).float(), a[0].grad.float()),
      "gxk", rel(xkT.grad.transpose(1, 2).float(), a[1].grad.float()), "gW", rel(Wb.grad.float(), a[2].grad.float()), "gb", rel(bb.grad.float(), a[3].grad.float()))
