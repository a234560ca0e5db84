"""Harness of the packed-fp16 CIN experiment (tools/experiments/cin_f16/cin_f16.hip, built by build.sh next to it; NOT
part of the product library): parity of each kernel against a float64 contraction on the device and timings at the
xDeepFM shape next to the shipped bf16 kernels of csrc/cin_mfma.hip.  Results: profiles/r04_cin_f16.md.

    bash tools/experiments/cin_f16/build.sh && python tools/experiments/cin_f16/cin16_dev.py [fwd] [bwd] [--B 65536]
    CIN16_LIB=<variant .so>: an ablation / staging variant built with build.sh -D... -o <name>
    CIN16_DATA=abs|relu|zeros: contents of the hidden-state operand (power-dependent clocks)
"""
import argparse
import os
import sys

import torch

import ctypes  # noqa: E402
from ctypes import c_int32, c_int64, c_size_t, c_void_p  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from torecsys_amd import _abi  # noqa: E402
from torecsys_amd._abi import ptr, stream_ptr  # noqa: E402
from torecsys_amd import functional as F_  # noqa: E402

_abi.load()                                       # libtrs_hip.so first: the experiment library resolves its helpers there
_P, _I32, _I64, _SZ = c_void_p, c_int32, c_int64, c_size_t
_SIG = {
    "trs_cin16_supported": (c_int32, [_I32, _I32, _I32, _I32]),
    "trs_cin16_fwd_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "trs_cin16_fwd": (c_int32, [_P, _P, _I32, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_cin16_bwd_data_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "trs_cin16_bwd_data": (c_int32, [_P, _P, _I32, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _P, _SZ, _P]),
}
_x = ctypes.CDLL(os.environ.get("CIN16_LIB") or os.path.join(HERE, "libcin16.so"))
for _n, (_r, _a) in _SIG.items():
    getattr(_x, _n).restype, getattr(_x, _n).argtypes = _r, _a


def size_query(name, *args):
    return int(getattr(_x, name)(*args)) if name in _SIG else _abi.size_query(name, *args)


def call(name, *args):
    if name not in _SIG:
        return _abi.call(name, *args)
    rc = getattr(_x, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {_abi.last_error()}")

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def make(B, N, H, C, E, tri, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x0 = torch.randn(B, N, E, generator=g).to(dev)
    if tri:
        xk = x0.clone()
    else:
        mode = os.environ.get("CIN16_DATA", "abs")
        xk = torch.randn(B, H, E, generator=g)
        xk = (xk.abs() if mode == "abs" else xk.clamp_min(0.) if mode == "relu" else xk * 0.).to(dev)   # post-ReLU-like
        if mode == "zeros":
            x0 = x0 * 0.
    W = (torch.randn(C, N, H, generator=g) * 0.05).to(dev)
    if tri:
        W = F_.cin_fold_symmetric(W.reshape(C, N * N), N).reshape(C, N, N)
    bias = torch.randn(C, generator=g).to(dev)
    return x0, xk, W, bias


def f16_inputs(x0, xk, H):
    B, N, E = x0.shape
    Hp = (H + 15) // 16 * 16
    x0h = x0.to(torch.float16).contiguous()
    xkT = torch.zeros(B, E, Hp, dtype=torch.float16, device=dev)
    xkT[:, :, :H] = xk.transpose(1, 2)
    return x0h, xkT


def run_fwd16(x0h, xkT, W, bias, N, H, C, tri, out_mul=None):
    B, _, E = x0h.shape
    yT = torch.empty(B, E, C, dtype=torch.bfloat16, device=dev)
    wsb = size_query("trs_cin16_fwd_workspace_bytes", N, H, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    W32 = W.reshape(C, N * H).float().contiguous()
    call("trs_cin16_fwd", ptr(x0h), ptr(xkT), xkT.shape[2], ptr(W32), ptr(bias), ptr(out_mul), B, N, H, C, E, int(tri),
         ptr(yT), ptr(ws), wsb, stream_ptr())
    return yT


def ref_fwd(x0h, xkT, W, bias, H):
    # float64 on the fp16-rounded operands
    x0 = x0h.double()
    xk = xkT[:, :, :H].double()                       # (B,E,H)
    Wd = W.to(torch.float16).double()
    y = torch.einsum("cnh,bne,beh->bec", Wd, x0, xk) + bias.double()
    return y


def check_fwd(Bc, N, H, C, E, tri):
    x0, xk, W, bias = make(Bc, N, H, C, E, tri, seed=1)
    x0h, xkT = f16_inputs(x0, xk, H)
    y = run_fwd16(x0h, xkT, W, bias, N, H, C, tri)
    torch.cuda.synchronize()
    r = ref_fwd(x0h, xkT, W, bias, H)
    err = float((y.double() - r).abs().max() / r.abs().max())
    print(f"fwd16 parity B={Bc} N={N} H={H} C={C} tri={tri}: max rel err {err:.3e} (bf16 output rounding ~4e-3)")
    mul = torch.tensor([0.25], device=dev)
    y2 = run_fwd16(x0h, xkT, W, bias, N, H, C, tri, out_mul=mul)
    r2 = (r - bias.double()) * 0.25 + bias.double()
    err2 = float((y2.double() - r2).abs().max() / r2.abs().max())
    print(f"      with out_mul=0.25: {err2:.3e}")
    return err < 8e-3 and err2 < 8e-3


def bench_fwd(B, N, H, C, E, tri):
    x0, xk, W, bias = make(B, N, H, C, E, tri)
    x0h, xkT = f16_inputs(x0, xk, H)
    ms = timeit(lambda: run_fwd16(x0h, xkT, W, bias, N, H, C, tri))
    flop = 2.0 * B * E * C * N * H
    print(f"fwd16 B={B} N={N} H={H} C={C} tri={tri}: {ms:.3f} ms  {flop / ms / 1e12:.3f} PFLOP/s algorithmic")
    # the bf16 kernel of cin_mfma.hip on the same shape
    ld0 = (N + 31) // 32 * 32
    x0T = torch.zeros(B, E, ld0, dtype=torch.bfloat16, device=dev)
    x0T[:, :, :N] = x0.transpose(1, 2)
    if tri:
        xkTb, ldk = x0T, ld0
    else:
        xkTb = xk.transpose(1, 2).contiguous().to(torch.bfloat16)
        ldk = H
    Wb = W.reshape(C, N * H).to(torch.bfloat16).contiguous()
    bb = bias.to(torch.bfloat16)
    yT = torch.empty(B, E, C, dtype=torch.bfloat16, device=dev)
    wsb = size_query("trs_cin_cl_workspace_bytes", N, H, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def old():
        call("trs_cin_cl_fwd", ptr(x0T), ld0, ptr(xkTb), ldk, ptr(Wb), ptr(bb), B, N, H, C, E, _abi.TRS_BF16, int(tri),
             ptr(yT), ptr(ws), wsb, stream_ptr())
    ms0 = timeit(old)
    print(f"   bf16 cin_cl_fwd: {ms0:.3f} ms  {flop / ms0 / 1e12:.3f} PFLOP/s")


def run_bwd16(x0h, xkT, gyT, W, N, H, C, tri, out_mul=None):
    B, _, E = x0h.shape
    ldo = (H + 31) // 32 * 32
    dx0 = torch.empty(B, N, E, dtype=torch.bfloat16, device=dev)
    dxkT = torch.empty(B, E, ldo, dtype=torch.bfloat16, device=dev)
    wsb = size_query("trs_cin16_bwd_data_workspace_bytes", N, H, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    W32 = W.reshape(C, N * H).float().contiguous()
    call("trs_cin16_bwd_data", ptr(x0h), ptr(xkT), xkT.shape[2], ptr(gyT), ptr(W32), ptr(out_mul), B, N, H, C, E, int(tri),
         ptr(dx0), ptr(dxkT), ldo, ptr(ws), wsb, stream_ptr())
    return dx0, dxkT


def f16_inputs32(x0, xk, H):
    B, N, E = x0.shape
    Hp = (H + 31) // 32 * 32
    x0h = x0.to(torch.float16).contiguous()
    xkT = torch.zeros(B, E, Hp, dtype=torch.float16, device=dev)
    xkT[:, :, :H] = xk.transpose(1, 2)
    return x0h, xkT


def check_bwd(Bc, N, H, C, E, tri):
    x0, xk, W, _ = make(Bc, N, H, C, E, tri, seed=2)
    x0h, xkT = f16_inputs32(x0, xk, H)
    g = torch.Generator(device="cpu").manual_seed(5)
    gyT = (torch.randn(Bc, E, C, generator=g) * 8.0).to(dev).to(torch.float16)
    mul = torch.tensor([0.125], device=dev)
    dx0, dxkT = run_bwd16(x0h, xkT, gyT, W, N, H, C, tri, out_mul=mul)
    torch.cuda.synchronize()
    Wd = W.to(torch.float16).double()
    gy = gyT.double() * 0.125
    r_dxk = torch.einsum("bec,cnh,bne->beh", gy, Wd, x0h.double())
    r_dx0 = torch.einsum("bec,cnh,beh->bne", gy, Wd, xkT[:, :, :H].double())
    e1 = float((dxkT[:, :, :H].double() - r_dxk).abs().max() / r_dxk.abs().max())
    e2 = float((dx0.double() - r_dx0).abs().max() / r_dx0.abs().max())
    pad = float(dxkT[:, :, H:].abs().max()) if dxkT.shape[2] > H else 0.0
    print(f"bwd16 parity B={Bc} N={N} H={H} C={C} tri={tri}: dxk {e1:.3e} dx0 {e2:.3e} pad {pad}")
    return e1 < 8e-3 and e2 < 8e-3 and pad == 0.0


def bench_bwd(B, N, H, C, E, tri):
    x0, xk, W, _ = make(B, N, H, C, E, tri)
    x0h, xkT = f16_inputs32(x0, xk, H)
    g = torch.Generator(device="cpu").manual_seed(5)
    gy = torch.randn(B, E, C, generator=g).to(dev)
    gyT = gy.to(torch.float16)
    ms = timeit(lambda: run_bwd16(x0h, xkT, gyT, W, N, H, C, tri))
    flop = 2.0 * B * E * C * N * H
    print(f"bwd16 B={B} N={N} H={H} C={C} tri={tri}: {ms:.3f} ms  {flop / ms / 1e12:.3f} PFLOP/s algorithmic")
    ld0 = (N + 31) // 32 * 32
    x0T = torch.zeros(B, E, ld0, dtype=torch.bfloat16, device=dev)
    x0T[:, :, :N] = x0.transpose(1, 2)
    if tri:
        xkTb, ldk = x0T, ld0
    else:
        xkTb = xk.transpose(1, 2).contiguous().to(torch.bfloat16)
        ldk = H
    Wb = W.reshape(C, N * H).to(torch.bfloat16).contiguous()
    gyb = gy.to(torch.bfloat16)
    ldo = (H + 31) // 32 * 32
    dx0T = torch.empty_like(x0T)
    dxk = None if tri else torch.empty(B, E, ldo, dtype=torch.bfloat16, device=dev)
    wsb = size_query("trs_cin_cl_bwd_data_workspace_bytes", N, H, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def old():
        call("trs_cin_cl_bwd_data", ptr(x0T), ld0, ptr(xkTb), ldk, ptr(gyb), ptr(Wb), B, N, H, C, E, _abi.TRS_BF16,
             int(tri), ptr(dx0T), ptr(dxk), ldo, ptr(ws), wsb, stream_ptr())
    ms0 = timeit(old)
    print(f"   bf16 cin_cl_bwd_data: {ms0:.3f} ms  {flop / ms0 / 1e12:.3f} PFLOP/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["fwd"])
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--check-B", type=int, default=512)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    ok = True
    if "fwd" in a.what and a.no_check:
        bench_fwd(a.B, 39, 128, 256, 64, False)
        bench_fwd(a.B, 39, 39, 256, 64, True)
    elif "fwd" in a.what:
        for (N, H, C, tri) in [(39, 128, 256, False), (39, 39, 256, True), (39, 39, 256, False), (10, 64, 128, False),
                               (39, 80, 256, False)]:
            ok &= check_fwd(a.check_B, N, H, C, 64, tri)
        ok &= check_fwd(1021, 39, 128, 256, 64, False)        # dead waves in the last round
        bench_fwd(a.B, 39, 128, 256, 64, False)
        bench_fwd(a.B, 39, 39, 256, 64, True)
    if "bwd" in a.what:
        if not a.no_check:
            for (N, H, C, tri) in [(39, 128, 256, False), (39, 39, 256, True), (39, 39, 256, False), (10, 64, 128, False),
                                   (39, 80, 256, False)]:
                ok &= check_bwd(a.check_B, N, H, C, 64, tri)
            ok &= check_bwd(1021, 39, 128, 256, 64, False)
        bench_bwd(a.B, 39, 128, 256, 64, False)
        bench_bwd(a.B, 39, 39, 256, 64, True)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
