"""Developer probe: split-K slice count for the per-field MLP of DCN (2 555 904 rows), GPU time by events."""
import torch
dev = torch.device("cuda:0"); bf = torch.bfloat16
M = 65536 * 39
def t(fn, n=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for (H, K) in ((512, 512), (512, 64), (64, 512)):
    g = torch.randn(M, H, device=dev, dtype=bf); x = torch.randn(M, K, device=dev, dtype=bf)
    for S in (39, 78, 156, 312, 624, 1248):
        a = t(lambda: torch.bmm(g.view(S, M // S, H).transpose(1, 2), x.view(S, M // S, K), out_dtype=torch.float32))
        b = t(lambda: torch.bmm(x.view(S, M // S, K).transpose(1, 2), g.view(S, M // S, H), out_dtype=torch.float32))
        print(f"out {H}x{K}  S={S:4d}:  g^T x %.0f us   x^T g %.0f us   (+ finish over %.0f MB)" % (a, b, S * H * K * 4 / 1e6))
    del g, x
