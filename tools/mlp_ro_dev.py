"""dev harness: row-owner fused MLP forward against a float reference, and timings (TRS_MLP_RO=1 vs 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torecsys_amd import functional as F_

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
dt = torch.bfloat16


def ref(x, Ws, bs):
    hs = []
    h = x.float()
    for l, (w, b) in enumerate(zip(Ws, bs)):
        h = h @ w.float().t() + b.float()
        if l + 1 < len(Ws):
            h = torch.relu(h).to(dt).float()
            hs.append(h)
    return h.to(dt), hs


def run(widths, rows, input_mask=False):
    Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dt).to(dev) for i, o in zip(widths[:-1], widths[1:])]
    bs = [(0.1 * torch.randn(o, generator=g)).to(dt).to(dev) for o in widths[1:]]
    x = torch.randn(rows, widths[0], generator=g).to(dt).to(dev)
    if input_mask:
        x = torch.relu(x)
    out = F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=input_mask)
    y, hidden = out[0], out[1]
    torch.cuda.synchronize()
    yr, hr = ref(x, Ws, bs)
    e = float((y.float() - yr.float()).abs().max() / yr.float().abs().max())
    eh = [float((h[:, :w].float() - r).abs().max() / r.abs().max()) for h, r, w in zip(hidden, hr, widths[1:])]
    pad = [float(h[:, w:].float().abs().max()) if h.shape[1] > w else 0.0 for h, w in zip(hidden, widths[1:])]
    print(f"widths {widths} rows {rows}: y rel err {e:.3e}, hidden {['%.2e' % v for v in eh]}, pad max {pad}", flush=True)
    if rows >= 65536:
        for _ in range(2):
            F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=input_mask)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(); F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=input_mask); t1.record(); torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1))
        fl = 2.0 * rows * sum(i * o for i, o in zip(widths[:-1], widths[1:]))
        t = sorted(ts)[len(ts) // 2]
        print(f"   fwd med {t:.3f} ms  {fl / t / 1e9:.1f} TFLOP/s", flush=True)


print("TRS_MLP_RO =", os.environ.get("TRS_MLP_RO"))
for rows in (256, 1000, 70000, 65536 * 39):
    run([64, 400, 400, 400, 64], rows)
if os.environ.get("RO_TAIL", "1") == "1":
    for rows in (300, 65536):
        run([416, 400, 400, 8], rows, input_mask=True)


def run_bwd(widths, rows, input_mask=False):
    Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dt).to(dev) for i, o in zip(widths[:-1], widths[1:])]
    bs = [(0.1 * torch.randn(o, generator=g)).to(dt).to(dev) for o in widths[1:]]
    x = torch.randn(rows, widths[0], generator=g).to(dt).to(dev)
    if input_mask:
        x = torch.relu(x)
    out = F_.fused_mlp_forward_raw(x, Ws, bs, input_mask=input_mask)
    y, hidden, masks = out[0], out[1], out[2]
    mask_in = out[3] if input_mask else None
    gy = torch.randn(rows, widths[-1], generator=g).to(dt).to(dev)
    gx, gz, gb, gb_in = F_.fused_mlp_backward_raw(gy, widths, Ws, masks, mask_in)
    torch.cuda.synchronize()
    # reference under the kernel's own masks (signs of the stored hidden activations)
    L = len(Ws)
    gcur = gy.float()
    errs = []
    gbr = [None] * L
    for l in range(L - 1, -1, -1):
        gbr[l] = gcur.sum(0)
        gprev = gcur @ Ws[l].float()
        if l > 0:
            gprev = gprev * (hidden[l - 1][:, :widths[l]].float() > 0)
            gprev = gprev.to(dt).float()
            e = float((gz[l - 1][:, :widths[l]].float() - gprev).abs().max() / gprev.abs().max())
            errs.append(e)
        elif input_mask:
            gprev = gprev * (x.float() > 0)
        gcur = gprev
    ex = float((gx.float() - gcur).abs().max() / gcur.abs().max())
    eb = [float((gb[l][:widths[l + 1]] - gbr[l]).abs().max() / gbr[l].abs().max()) for l in range(L)]
    s = f"bwd widths {widths} rows {rows}: gx rel err {ex:.3e}, gz {['%.2e' % v for v in errs[::-1]]}, gb {['%.2e' % v for v in eb]}"
    if input_mask:
        r = gcur.sum(0)
        s += f", gb_in {float((gb_in[:widths[0]] - r).abs().max() / r.abs().max()):.2e}"
    print(s, flush=True)
    if rows >= 65536:
        for _ in range(2):
            F_.fused_mlp_backward_raw(gy, widths, Ws, masks, mask_in)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(); F_.fused_mlp_backward_raw(gy, widths, Ws, masks, mask_in); t1.record(); torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1))
        fl = 2.0 * rows * sum(i * o for i, o in zip(widths[:-1], widths[1:]))
        t = sorted(ts)[len(ts) // 2]
        print(f"   bwd med {t:.3f} ms  {fl / t / 1e9:.1f} TFLOP/s", flush=True)


for rows in (256, 1000, 70000, 65536 * 39):
    run_bwd([64, 400, 400, 400, 64], rows)
if os.environ.get("RO_TAIL", "1") == "1":
    for rows in (300, 65536):
        run_bwd([416, 400, 400, 8], rows, input_mask=True)
