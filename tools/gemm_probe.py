"""Developer probe: hipBLASLt bf16 GEMM rates for the DeepFM MLP shapes and padded variants."""
import torch, time
dev = torch.device("cuda:0")
M, K = 65536, 2496
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
for K2 in (2496, 2560):
    xx = torch.randn(M, K2, device=dev, dtype=torch.bfloat16)
    for N in (400, 384, 448, 512):
        w = torch.randn(N, K2, device=dev, dtype=torch.bfloat16)
        g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        f = 2.0 * M * K2 * N
        a = t(lambda: xx @ w.t()); b = t(lambda: g @ w); c = t(lambda: g.t() @ xx)
        print(f"K={K2} N={N}: fwd {a*1e6:7.1f} us {f/a/1e12:6.0f} TF/s | dgrad {b*1e6:7.1f} us {f/b/1e12:6.0f} | wgrad {c*1e6:7.1f} us {f/c/1e12:6.0f}", flush=True)
for N in (400, 512):
    h = torch.randn(M, N, device=dev, dtype=torch.bfloat16); w = torch.randn(N, N, device=dev, dtype=torch.bfloat16)
    f = 2.0 * M * N * N
    a = t(lambda: h @ w.t()); b = t(lambda: h @ w); c = t(lambda: h.t() @ h)
    print(f"hidden {N}x{N}: fwd {a*1e6:7.1f} us {f/a/1e12:6.0f} TF/s | dgrad {b*1e6:7.1f} {f/b/1e12:6.0f} | wgrad {c*1e6:7.1f} {f/c/1e12:6.0f}", flush=True)
