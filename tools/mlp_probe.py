"""Developer probe: time the DeepFM MLP (outside the hand-written path) fwd+bwd under different GEMM settings."""
import os, sys, time, torch, torch.nn as nn
dev = torch.device("cuda:0")
B, K = 65536, 2496
def build():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(K, 400), nn.ReLU(), nn.Linear(400, 400), nn.ReLU(), nn.Linear(400, 400), nn.ReLU(),
                         nn.Linear(400, 1)).to(dev).bfloat16()
x = torch.randn(B, K, device=dev, dtype=torch.bfloat16, requires_grad=True)
def run(m, n=10):
    for _ in range(3):
        m(x).sum().backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        m(x).sum().backward()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
m = build()
print("default           ms/iter", round(run(m), 3), flush=True)
for lib in ("cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        print(f"preferred {lib:8s} ms/iter", round(run(m), 3), flush=True)
    except Exception as e:
        print(lib, "failed", e)
if "tune" in sys.argv:
    torch.backends.cuda.preferred_blas_library("cublaslt")
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.set_max_tuning_duration(50)
    torch.cuda.tunable.set_max_tuning_iterations(20)
    t = time.perf_counter()
    print("tunableop (tuning) ms/iter", round(run(m, 2), 3), "tuning took", round(time.perf_counter() - t, 1), "s", flush=True)
    print("tunableop (tuned)  ms/iter", round(run(m), 3), flush=True)
