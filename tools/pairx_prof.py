"""Developer probe: one forward+backward of each N3 layer at B=8192 (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torecsys_amd.layers import AttentionalFactorizationMachineLayer, BilinearInteractionLayer, OuterProductNetworkLayer
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, N, E = 8192, 39, 64
which = sys.argv[1] if len(sys.argv) > 1 else "bil_all"
lay = {"bil_all": lambda: BilinearInteractionLayer(E, N, "all"), "bil_each": lambda: BilinearInteractionLayer(E, N, "each"),
       "opn_vec": lambda: OuterProductNetworkLayer(E, N, "vec"), "opn_mat": lambda: OuterProductNetworkLayer(E, N, "mat"),
       "afm": lambda: AttentionalFactorizationMachineLayer(E, N, 64, 0.0)}[which]().to(dev).to(dt)
x = (0.5 * torch.randn(B, N, E)).to(dt).to(dev).requires_grad_()
for _ in range(3):
    y = lay(x); y = y[0] if isinstance(y, tuple) else y
    y.rename(None).float().sum().backward()
torch.cuda.synchronize()
