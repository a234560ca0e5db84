#!/usr/bin/env python3
"""Kernel micro-benchmarks (developer tool): time each C-ABI kernel at the BASELINE shape with HIP events
on the launch stream and print algorithmic GB/s.  python tools/kbench.py [--dtype bf16] [--what a,b]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torecsys_amd import functional as F_  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--N", type=int, default=39)
    ap.add_argument("--E", type=int, default=64)
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--zipf", action="store_true")
    ap.add_argument("--what", default="all")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    s = 2 if dt == torch.bfloat16 else 4
    dev = torch.device("cuda:0")
    B, N, E, V = a.B, a.N, a.E, a.V
    g = torch.Generator().manual_seed(1234)
    per = V // N
    fs = [per] * (N - 1) + [V - per * (N - 1)]
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.tensor(fs), 0)[:-1]]).to(dev)
    if a.zipf:
        cols = []
        for f in fs:
            r = torch.rand(B, 1, generator=g, dtype=torch.float64)
            # Zipf(1.05)-like via inverse power transform
            cols.append(((f ** r - 1).clamp_(0, f - 1)).long())
        idx = torch.cat(cols, 1).to(dev)
    else:
        idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev)
    w = torch.randn(V, E, generator=g).to(dt).to(dev)
    w1 = torch.randn(V, 1, generator=g).to(dt).to(dev)
    what = a.what.split(",")

    def want(k):
        return "all" in what or k in what

    def report(name, t, nbytes):
        print(f"{name:34s} med {t[0]*1e6:9.1f} us  min {t[1]*1e6:9.1f} us  {nbytes/t[0]/1e9:8.1f} GB/s (alg) "
              f"{nbytes/t[0]/8e12*100:5.1f}% of 8TB/s", flush=True)

    rd = B * N * (8 + E * s)
    if want("gather"):
        t = timeit(lambda: F_._GatherRows.apply(w, idx, off, None))
        report("gather_rows", t, rd + B * N * E * s)
    if want("embed_fm"):
        t = timeit(lambda: F_._EmbedFM.apply(w, idx, off, None, False))
        report("embed_fm (fm only)", t, rd + B * E * s + B * E * 4)
        t = timeit(lambda: F_._EmbedFM.apply(w, idx, off, None, True))
        report("embed_fm (+emb block)", t, rd + B * N * E * s + B * E * s + B * E * 4)
        t = timeit(lambda: F_._EmbedFM.apply(w, idx, off, w1, True))
        report("embed_fm (+emb +first)", t, rd + B * N * s + B * N * E * s + B * E * s + B * E * 4)
        t = timeit(lambda: F_._EmbedFM.apply(w, idx, off, w1, False))
        report("embed_fm (fm+first, no block)", t, rd + B * N * s + B * E * s + B * E * 4)
    if want("fm"):
        x = torch.randn(B, N, E, generator=g).to(dt).to(dev)
        t = timeit(lambda: F_._FMLayer.apply(x))
        report("fm_fwd (block)", t, B * N * E * s + B * E * s + B * E * 4)
    if want("csr"):
        def csr():
            F_.clear_caches()
            return F_.row_buckets(idx, off, V)
        t = timeit(csr)
        report("csr_build", t, B * N * (8 + 4 + 4 + 4) + 2 * (V + 1) * 4 * 2)
    if want("scatter"):
        rb = F_.row_buckets(idx, off, V)
        ge = torch.randn(B, N, E, generator=g).to(dt).to(dev)
        gf = torch.randn(B, E, generator=g).to(dt).to(dev)
        S = torch.randn(B, E, generator=g).to(dev)
        t = timeit(lambda: F_.scatter_rows(rb, w, g_rows=ge))
        report("scatter_rows (g_rows)", t, B * N * (E * s + 4) + V * E * s + V * 4)
        t = timeit(lambda: F_.scatter_rows(rb, w, g_rows=ge, g_bcast=gf, fm_sum=S))
        report("scatter_rows (g_rows + fm)", t, B * N * (E * s + 4) + V * E * s * 2 + V * 4)
        g1 = gf[:, :1].contiguous()
        t = timeit(lambda: F_.scatter_rows(rb, w, g_rows=ge, g_bcast=g1, fm_sum=S))
        report("scatter_rows (g_rows + fm, g constant along E)", t, B * N * (E * s + 4) + V * E * s * 2 + V * 4)
        t = timeit(lambda: F_.scatter_rows(rb, w1, g_bcast=gf[:, :1].contiguous()))
        report("scatter_rows (first-order E=1)", t, B * N * (4) + V * s + V * 4)
    if want("cross"):
        L = 6
        x = (0.5 * torch.randn(B, N, E, generator=g)).to(dt).to(dev).requires_grad_()
        W = (torch.randn(L, E, E, generator=g) / E ** 0.5).to(dt).to(dev).requires_grad_()
        bb = (0.1 * torch.randn(L, E, generator=g)).to(dt).to(dev).requires_grad_()
        t = timeit(lambda: F_._Cross.apply(x.detach(), W.detach(), bb.detach(), True), iters=5, warm=1)
        report(f"cross_fwd L={L}", t, 2 * B * N * E * s)
        print(f"    -> {2*B*N*E*E*L/t[0]/1e12:.1f} TFLOP/s")
        y = F_._Cross.apply(x, W, bb, True)
        gy = torch.randn_like(y)
        t = timeit(lambda: torch.autograd.grad(y, (x, W, bb), gy, retain_graph=True), iters=3, warm=1)
        report(f"cross_bwd L={L}", t, 3 * B * N * E * s)
    if want("ipn"):
        x = torch.randn(B, N, E, generator=g).to(dt).to(dev).requires_grad_()
        t = timeit(lambda: F_._PairDot.apply(x.detach()), iters=10)
        report("pair_dot_fwd", t, B * N * E * s + B * (N * (N - 1) // 2) * s)
        y = F_._PairDot.apply(x)
        gy = torch.randn_like(y)
        t = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True), iters=10)
        report("pair_dot_bwd", t, 2 * B * N * E * s + B * (N * (N - 1) // 2) * s)
        Pn = N * (N - 1) // 2
        with torch.no_grad():
            t = timeit(lambda: F_.gather_rows(w, idx, off), iters=10)
            t2 = timeit(lambda: F_.embed_ipn(w, idx, off, want_emb=True), iters=10)
            report("gather_rows alone (for comparison)", t, B * N * (8 + 2 * E * s))
            report("embed_ipn (lookup + inner products, block written)", t2, B * N * (8 + 2 * E * s) + B * Pn * s)
            t3 = timeit(lambda: F_.embed_ipn(w, idx, off, want_emb=False), iters=10)
            report("embed_ipn (no block: inference)", t3, B * N * (8 + E * s) + B * Pn * s)
    if want("pairx"):
        from torecsys_amd.layers import (AttentionalFactorizationMachineLayer, BilinearInteractionLayer,
                                         OuterProductNetworkLayer)
        P = N * (N - 1) // 2
        Bp = a.B // 8 if a.B >= 8192 else a.B          # the (B,NC2,E) outputs are 6.2 GB at B = 65536
        x = (0.5 * torch.randn(Bp, N, E, generator=g)).to(dt).to(dev).requires_grad_()
        fl_bil = 2.0 * Bp * P * E * E

        def run(name, lay, out_elems, flops):
            lay = lay.to(dev).to(dt)
            f = lambda: lay(x.detach())
            t = timeit(f, iters=3, warm=1)
            yb = out_elems * s + Bp * N * E * s
            print(f"{name:34s} fwd med {t[0]*1e6:9.1f} us  {yb/t[0]/1e9:7.1f} GB/s (alg)  {flops/t[0]/1e12:7.1f} TFLOP/s",
                  flush=True)
            y = lay(x)
            y = y[0] if isinstance(y, tuple) else y
            gy = torch.randn_like(y.rename(None))
            ins = (x,) + tuple(lay.parameters())
            t = timeit(lambda: torch.autograd.grad(y.rename(None), ins, gy, retain_graph=True), iters=3, warm=1)
            print(f"{'':34s} bwd med {t[0]*1e6:9.1f} us  {2*flops/t[0]/1e12:7.1f} TFLOP/s", flush=True)

        print(f"pair layers at B={Bp} N={N} E={E}")
        run("opn vec", OuterProductNetworkLayer(E, N, "vec"), Bp * P, 3.0 * Bp * P * E)
        run("opn num", OuterProductNetworkLayer(E, N, "num"), Bp * P, 2.0 * Bp * P * E)
        run("opn mat", OuterProductNetworkLayer(E, N, "mat"), Bp * P, fl_bil)
        run("bilinear all", BilinearInteractionLayer(E, N, "all"), Bp * P * E, 2.0 * Bp * N * E * E)
        run("bilinear each", BilinearInteractionLayer(E, N, "each"), Bp * P * E, fl_bil)
        run("afm A=64", AttentionalFactorizationMachineLayer(E, N, 64, 0.0), Bp * E + Bp * P, 2.0 * Bp * P * E * 64)
    if want("cin"):
        Bc = a.B // 8 if a.B >= 8192 else a.B
        import os as _os
        only = _os.environ.get("TRS_KB_CIN_H")
        for (H, C) in ((39, 256), (128, 256)):
            if only and int(only) != H:
                continue
            ld0 = ((N + 31) // 32) * 32
            x0T = torch.zeros(Bc, E, ld0, dtype=dt, device=dev)
            x0T[:, :, :N] = (0.5 * torch.randn(Bc, E, N, generator=g)).to(dt).to(dev)
            xkT = x0T if H == N else (0.5 * torch.randn(Bc, E, H, generator=g)).to(dt).to(dev)
            W = (torch.randn(C, N * H, generator=g) / (N * H) ** 0.5).to(dt).to(dev)
            bias = torch.zeros(C, dtype=dt, device=dev)
            t = timeit(lambda: F_._CINContractCL.apply(x0T, xkT, W, bias, N, H), iters=5, warm=1)
            fl = 2.0 * Bc * E * C * N * (H + 1)
            print(f"cin_cl_fwd B={Bc} N={N} H={H} C={C}: med {t[0]*1e3:.3f} ms  {fl/t[0]/1e12:.1f} TFLOP/s "
                  f"({fl/t[0]/2.5e15*100:.1f}% of 2.5 PF)", flush=True)
            x0r, xkr, Wr = x0T.clone().requires_grad_(), (None if H == N else xkT.clone().requires_grad_()), W.clone().requires_grad_()
            yT = F_._CINContractCL.apply(x0r, x0r if H == N else xkr, Wr, bias, N, H)
            gy = torch.randn_like(yT)
            ins = (x0r, Wr) if H == N else (x0r, xkr, Wr)
            t = timeit(lambda: torch.autograd.grad(yT, ins, gy, retain_graph=True), iters=3, warm=1)
            print(f"cin_cl_bwd (data+dW+transposes): med {t[0]*1e3:.3f} ms  {2*fl/t[0]/1e12:.1f} TFLOP/s", flush=True)
            t = timeit(lambda: torch.autograd.grad(yT, ins[:-1], gy, retain_graph=True), iters=5, warm=1)
            print(f"cin_cl_bwd data only H={H}: med {t[0]*1e3:.3f} ms  {fl/t[0]/1e12:.1f} TFLOP/s", flush=True)
            t = timeit(lambda: torch.autograd.grad(yT, ins[-1:], gy, retain_graph=True), iters=5, warm=1)
            print(f"cin_cl_bwd dW only (+transposes) H={H}: med {t[0]*1e3:.3f} ms  {fl/t[0]/1e12:.1f} TFLOP/s", flush=True)
    if want("mlp"):
        C = 400
        gy = torch.randn(B, C, generator=g).to(dt).to(dev)
        yy = torch.relu(torch.randn(B, C, generator=g)).to(dt).to(dev)
        t = timeit(lambda: F_.relu_bwd_bias(gy, yy))
        report("relu_bwd_bias (fused)", t, 3 * B * C * s)
        def aten():
            gz = torch.ops.aten.threshold_backward(gy, yy, 0)
            return gz, gz.sum(0)
        t = timeit(aten)
        report("ATen threshold_backward + sum", t, 4 * B * C * s)
    if want("mlpf"):
        rows_ = B * N
        widths = [64, 400, 400, 400, 64]
        Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dt).to(dev).requires_grad_() for i, o in zip(widths[:-1], widths[1:])]
        bs = [(0.1 * torch.randn(o, generator=g)).to(dt).to(dev).requires_grad_() for o in widths[1:]]
        xx = torch.randn(rows_, 64, generator=g).to(dt).to(dev).requires_grad_()
        fl = 2.0 * rows_ * sum(i * o for i, o in zip(widths[:-1], widths[1:]))
        t = timeit(lambda: F_.fused_mlp(xx.detach(), [w.detach() for w in Ws], [b_.detach() for b_ in bs]), iters=5, warm=1)
        print(f"fused MLP fwd  rows={rows_} {widths}: med {t[0]*1e3:.3f} ms  {fl/t[0]/1e12:.1f} TFLOP/s", flush=True)
        y = F_.fused_mlp(xx, Ws, bs)
        gy = torch.randn_like(y)
        t = timeit(lambda: torch.autograd.grad(y, (xx,), gy, retain_graph=True), iters=5, warm=1)
        print(f"fused MLP bwd (data only): med {t[0]*1e3:.3f} ms  {fl/t[0]/1e12:.1f} TFLOP/s", flush=True)
        t = timeit(lambda: torch.autograd.grad(y, (xx, *Ws, *bs), gy, retain_graph=True), iters=3, warm=1)
        print(f"fused MLP bwd (data + weight gradients): med {t[0]*1e3:.3f} ms  {2*fl/t[0]/1e12:.1f} TFLOP/s", flush=True)
    if want("mlpf") or want("wgrad"):
        # dW = g^T x over the rows: the hand-written kernel against the batched library GEMM it replaces
        for rows_, M_, N_ in [(B * N, 416, 416), (B * N, 416, 64), (B * N, 64, 416), (B, 416, 416), (B, 416, 2496)]:
            gg = torch.randn(rows_, M_, generator=g).to(dt).to(dev)
            xx2 = torch.randn(rows_, N_, generator=g).to(dt).to(dev)
            fl = 2.0 * rows_ * M_ * N_
            by = (M_ + N_) * 2.0 * rows_
            for on in (True, False):
                F_.WGRAD_ROWS = on
                t = timeit(lambda: F_._wgrad_rows(gg, xx2, M_, N_, dt), iters=5, warm=2)
                print(f"wgrad {'trs_wgrad_rows' if on else 'bmm split-K  '} rows={rows_} {M_}x{N_}: med {t[0]*1e6:8.1f} us  "
                      f"{fl/t[0]/1e12:6.1f} TFLOP/s  {by/t[0]/1e9:7.1f} GB/s (operands once)", flush=True)
            F_.WGRAD_ROWS = True
            del gg, xx2
    if "ffm" in what:         # (not part of "all": 5 GB of tables + 12.8 GB + 6.2 GB tensors at the BASELINE shape)
        # the fused field-aware lookup + FFM reads B*N*8 + B*N*N*E*s (12.8 GB at S) and writes B*NC2*E*s (6.2 GB)
        P = N * (N - 1) // 2
        tabs = [(torch.rand(V, E, generator=g) - 0.5).to(dt).to(dev) for _ in range(N)]
        rd_fa = B * N * 8 + B * N * N * E * s
        with torch.no_grad():
            t = timeit(lambda: F_.fa_gather_rows(tabs, idx, off), iters=5, warm=1)
            report("fa_gather_rows (B,N*N,E)", t, rd_fa + B * N * N * E * s)
            xfa = F_.fa_gather_rows(tabs, idx, off)
            t = timeit(lambda: F_.ffm_layer(xfa, N), iters=5, warm=1)
            report("ffm_fwd on the materialised block", t, 2 * B * P * E * s + B * P * E * s)
            t = timeit(lambda: F_.ffm_fused(tabs, idx, off), iters=5, warm=1)
            report("ffm_fused_fwd (lookup + FFM)", t, rd_fa + B * P * E * s)
            print(f"    -> SURVEY 8d bytes {(rd_fa + B * P * E * s) / 1e9:.2f} GB; 8 TB/s floor "
                  f"{(rd_fa + B * P * E * s) / 8e12 * 1e3:.2f} ms", flush=True)
        xr = xfa.detach().requires_grad_()
        y = F_.ffm_layer(xr, N)
        gy = torch.randn(B, P, E, dtype=dt, device=dev)
        t = timeit(lambda: torch.autograd.grad(y, xr, gy, retain_graph=True), iters=3, warm=1)
        report("ffm_bwd on the materialised block", t, 3 * B * P * E * s + B * N * N * E * s)
        del y, xr, xfa
        tr = [w_.requires_grad_() for w_ in tabs]
        y = F_.ffm_fused(tr, idx, off)
        t = timeit(lambda: torch.autograd.grad(y, tr, gy, retain_graph=True), iters=3, warm=1)
        report("ffm_fused_bwd (all N table grads)", t, 2 * B * N * N * E * s // 1 + N * V * E * s)
        del y, gy, tr, tabs
    if want("copy"):
        x = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        t = timeit(lambda: y.copy_(x))
        report("torch copy 512MB (r+w)", t, 2 * x.numel() * 4)


if __name__ == "__main__":
    main()
