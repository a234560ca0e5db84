"""Developer probe: host enqueue cost of the MLP GEMM calls (with / without TunableOp: run twice)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--tunable" in sys.argv:
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1" if "--tuning" in sys.argv else "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(ROOT, "torecsys_amd", "tuning", "tunableop_results.csv")
import torch
dev = torch.device("cuda:0")
B = 4096        # small rows: the GPU never becomes the limit, the loop measures the host
x = torch.randn(B, 2496, device=dev, dtype=torch.bfloat16)
W1 = torch.randn(512, 2496, device=dev, dtype=torch.bfloat16); b1 = torch.randn(512, device=dev, dtype=torch.bfloat16)
h = torch.randn(B, 512, device=dev, dtype=torch.bfloat16)
W2 = torch.randn(512, 512, device=dev, dtype=torch.bfloat16)
def bench(name, fn, n=2000):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn()
        if i % 200 == 199: torch.cuda.synchronize()
    el = time.perf_counter() - t
    torch.cuda.synchronize()
    print("%-40s %.2f us" % (name, el / n * 1e6))
bench("_addmm_activation (B,2496)x(2496,512)", lambda: torch._addmm_activation(b1, x, W1.t(), use_gelu=False))
bench("_addmm_activation (B,512)x(512,512)", lambda: torch._addmm_activation(b1, h, W2.t(), use_gelu=False))
bench("mm (B,512)x(512,512)", lambda: h @ W2)
bench("mm (B,512)x(512,2496)", lambda: h @ W1)
hs = h.view(2, B // 2, 512)
bench("bmm split-K wgrad", lambda: torch.bmm(hs.transpose(1, 2), hs, out_dtype=torch.float32))
bench("relu (B,512)", lambda: torch.relu(h))
bench("empty", lambda: torch.empty(B, 512, device=dev, dtype=torch.bfloat16))
