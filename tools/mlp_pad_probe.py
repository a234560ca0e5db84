"""Developer probe: per-op times of the DeepFM MLP fwd+bwd with hidden width 400 vs zero-padded 512, TunableOp tuned.
Ops mirror _LinearSplitK: addmm(+relu epilogue), dgrad mm, split-K bmm wgrad."""
import os, sys, time
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tune_probe.csv")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "30")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "10")
import torch
dev = torch.device("cuda:0")
M, K = 65536, 2496
bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
x = torch.randn(M, K, device=dev, dtype=bf)
for H in (400, 512):
    w1 = torch.randn(H, K, device=dev, dtype=bf) * 0.02; b1 = torch.zeros(H, device=dev, dtype=bf)
    w2 = torch.randn(H, H, device=dev, dtype=bf) * 0.02
    h = torch.randn(M, H, device=dev, dtype=bf); g = torch.randn(M, H, device=dev, dtype=bf)
    S = 32
    r = {}
    r["L1 fwd addmm"] = t(lambda: torch.addmm(b1, x, w1.t()))
    r["L1 fwd addmm+relu epi"] = t(lambda: torch._addmm_activation(b1, x, w1.t(), use_gelu=False))
    r["L1 dgrad"] = t(lambda: g @ w1)
    r["L1 wgrad splitK"] = t(lambda: torch.bmm(g.view(S, M // S, H).transpose(1, 2), x.view(S, M // S, K), out_dtype=torch.float32).sum(0))
    r["L1 wgrad mm"] = t(lambda: g.t() @ x)
    r["L2 fwd addmm"] = t(lambda: torch.addmm(b1, h, w2.t()))
    r["L2 fwd addmm+relu epi"] = t(lambda: torch._addmm_activation(b1, h, w2.t(), use_gelu=False))
    r["L2 dgrad"] = t(lambda: g @ w2)
    r["L2 wgrad splitK"] = t(lambda: torch.bmm(g.view(S, M // S, H).transpose(1, 2), h.view(S, M // S, H), out_dtype=torch.float32).sum(0))
    r["relu_ inplace"] = t(lambda: torch.relu_(h))
    print(f"--- hidden {H}")
    for k, v in r.items():
        print(f"{k:26s} {v:8.1f} us", flush=True)
    tot = min(r["L1 fwd addmm"] + r["relu_ inplace"], r["L1 fwd addmm+relu epi"]) + r["L1 dgrad"] + min(r["L1 wgrad splitK"], r["L1 wgrad mm"]) \
        + 2 * (min(r["L2 fwd addmm"] + r["relu_ inplace"], r["L2 fwd addmm+relu epi"]) + r["L2 dgrad"] + r["L2 wgrad splitK"])
    print(f"best-of total (3 layers) {tot:8.1f} us", flush=True)
