"""Developer probe: first-layer weight gradient, split count S and operand order, GPU time by events (TunableOp tuning on)."""
import os
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1"); os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tune_probe3.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "60"); os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "30")
import torch
dev = torch.device("cuda:0"); bf = torch.bfloat16
M, K, H = 65536, 2496, 512
x = torch.randn(M, K, device=dev, dtype=bf); g = torch.randn(M, H, device=dev, dtype=bf)
def t(fn, n=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for S in (2, 4, 8, 16, 32, 64):
    print(f"g^T x  (S,512,2496) fp32  S={S:2d}: %.1f us" % t(lambda: torch.bmm(g.view(S, M // S, H).transpose(1, 2), x.view(S, M // S, K), out_dtype=torch.float32)))
for S in (4, 8, 16, 32):
    print(f"x^T g  (S,2496,512) fp32  S={S:2d}: %.1f us" % t(lambda: torch.bmm(x.view(S, M // S, K).transpose(1, 2), g.view(S, M // S, H), out_dtype=torch.float32)))
print("single g^T x bf16 out: %.1f us" % t(lambda: g.t() @ x))
print("single x^T g bf16 out: %.1f us" % t(lambda: x.t() @ g))
print("--- 512 x 512 layers")
K2 = 512
x2 = torch.randn(M, K2, device=dev, dtype=bf)
for S in (8, 16, 32, 64):
    print(f"g^T x2 (S,512,512) fp32  S={S:2d}: %.1f us" % t(lambda: torch.bmm(g.view(S, M // S, H).transpose(1, 2), x2.view(S, M // S, K2), out_dtype=torch.float32)))
print("--- repeat of the candidates (fresh tensors)")
x = torch.randn(M, K, device=dev, dtype=bf); g = torch.randn(M, H, device=dev, dtype=bf)
for S in (16, 32):
    print(f"x^T g  S={S:2d}: %.1f us" % t(lambda: torch.bmm(x.view(S, M // S, K).transpose(1, 2), g.view(S, M // S, H), out_dtype=torch.float32)))
    print(f"g^T x  S={S:2d}: %.1f us" % t(lambda: torch.bmm(g.view(S, M // S, H).transpose(1, 2), x.view(S, M // S, K), out_dtype=torch.float32)))
