"""Developer probe: wgrad GEMM (g^T x, K = batch) as one GEMM vs manual split-K through bmm."""
import torch, time
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
B = 65536
for (O_, I_) in ((400, 400), (400, 2496), (1, 400)):
    g = torch.randn(B, O_, device=dev, dtype=torch.bfloat16)
    x = torch.randn(B, I_, device=dev, dtype=torch.bfloat16)
    ref = (g.float().t() @ x.float())
    t0 = timeit(lambda: g.t() @ x)
    print(f"wgrad {O_}x{I_}: mm {t0:.1f} us", end="")
    for S in (16, 32, 64, 128, 256):
        g3, x3 = g.view(S, B // S, O_), x.view(S, B // S, I_)
        f = lambda: torch.bmm(g3.transpose(1, 2), x3).float().sum(0)
        t1 = timeit(f)
        err = float((f() - ref).abs().max() / ref.abs().max())
        print(f" | S={S}: {t1:.1f} us (err {err:.1e})", end="")
        try:
            f2 = lambda: torch.bmm(g3.transpose(1, 2), x3, out_dtype=torch.float32).sum(0)
            t2 = timeit(f2)
            print(f" f32out {t2:.1f}", end="")
        except Exception as e:
            pass
    print()
