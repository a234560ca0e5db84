"""Developer probe: first-layer input gradient g (65536x512) W (512x2496): NN on W vs TN on a transposed copy."""
import os
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1"); os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tune_probe4.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "60"); os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "30")
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0"); bf = torch.bfloat16
M = 65536
def t(fn, n=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for (H, K) in ((512, 2496), (512, 512)):
    g = torch.randn(M, H, device=dev, dtype=bf); W = torch.randn(H, K, device=dev, dtype=bf); Wt = W.t().contiguous()
    print(f"H={H} K={K}:  g @ W (NN) %.1f us   F.linear(g, Wt) (TN) %.1f us" % (t(lambda: g @ W), t(lambda: F.linear(g, Wt))))
    x = torch.randn(M, K, device=dev, dtype=bf); b = torch.randn(H, device=dev, dtype=bf)
    print(f"   fwd addmm_activation %.1f us" % t(lambda: torch._addmm_activation(b, x, W.t().contiguous().t() if False else Wt.t().contiguous().t(), use_gelu=False)))
