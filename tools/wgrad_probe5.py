"""First-layer weight gradient of the DeepFM deep branch (dW1 = x^T g, K = 65 536 rows): the split-K batched form the
layers use against single library GEMMs with fp32 output (hipBLASLt's own split-K / stream-K solutions, tuned by TunableOp).
usage (GPU box): python tools/wgrad_probe5.py"""
import os, sys, tempfile
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "200")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "50")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tempfile.gettempdir(), "probe5_tunable.csv"))
import torch
dev = torch.device("cuda:0")
rows, K1, H = 65536, 2496, int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randn(rows, K1, device=dev).bfloat16()
g = torch.randn(rows, H, device=dev).bfloat16()


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for St in (8, 16, 32):
    us = timeit(lambda: torch.bmm(x.view(St, rows // St, -1).transpose(1, 2), g.view(St, rows // St, -1), out_dtype=torch.float32))
    print(f"bmm x^T g  {St:3d} slices (fp32 partials)      {us:8.1f} us")
for St in (16, 32):
    us = timeit(lambda: torch.bmm(g.view(St, rows // St, -1).transpose(1, 2), x.view(St, rows // St, -1), out_dtype=torch.float32))
    print(f"bmm g^T x  {St:3d} slices (fp32 partials)      {us:8.1f} us")
for name, fn in (("mm  x^T g  single GEMM, fp32 out", lambda: torch.mm(x.t(), g, out_dtype=torch.float32)),
                 ("mm  g^T x  single GEMM, fp32 out", lambda: torch.mm(g.t(), x, out_dtype=torch.float32)),
                 ("mm  x^T g  single GEMM, bf16 out", lambda: torch.mm(x.t(), g)),
                 ("mm  g^T x  single GEMM, bf16 out", lambda: torch.mm(g.t(), x))):
    try:
        print(f"{name}         {timeit(fn):8.1f} us")
    except Exception as exc:  # noqa: BLE001
        print(f"{name}         failed: {type(exc).__name__}: {str(exc)[:80]}")
