"""Developer probe: first-layer weight gradient formulations under TunableOp (M=65536 rows, 512 x 2496 output)."""
import os, time
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1"); os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tune_probe2.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "60"); os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "30")
import torch
dev = torch.device("cuda:0"); bf = torch.bfloat16
M, K, H = 65536, 2496, 512
x = torch.randn(M, K, device=dev, dtype=bf); g = torch.randn(M, H, device=dev, dtype=bf)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
print("g.t() @ x        ", round(t(lambda: g.t() @ x), 1))
print("(x.t() @ g).t()  ", round(t(lambda: (x.t() @ g)), 1))
for S in (16, 32, 64):
    print(f"split-K bmm^T S={S:2d}", round(t(lambda: torch.bmm(x.view(S, M // S, K).transpose(1, 2), g.view(S, M // S, H), out_dtype=torch.float32).sum(0)), 1))

for S in (16, 32):
    print(f"split-K bmm^T bf16-out S={S:2d}", round(t(lambda: torch.bmm(x.view(S, M // S, K).transpose(1, 2), g.view(S, M // S, H)).float().sum(0)), 1))
