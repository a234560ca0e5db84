#!/usr/bin/env python3
"""Developer tool: per-phase time stamps of cross_mfma_bwd3 (workgroup 0).  Builds a -DB3_TRACE copy of the library
under gpurun_out/, runs one backward at the BASELINE shape and prints the cycle deltas of a few groups.
    python tools/cross_trace.py        (on the GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = "/tmp/libtrs_trace.so"
EXTRA = sys.argv[1:]            # e.g. -DB3_NO_DW: ablations
from torecsys_amd import build as B  # noqa: E402
objs = []
for src in B.sources():
    o = os.path.join("/tmp", src[:-4] + ".trace.o")
    cmd = [B.HIPCC, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-DB3_TRACE", *EXTRA, "-c", os.path.join(B.CSRC, src), "-o", o]
    if src == "cross_mfma.hip" or not os.path.exists(o):
        subprocess.run(cmd, check=True)
    objs.append(o)
subprocess.run(["g++", "-shared", "-o", out, *objs, "-L" + B.torch_lib_dir(), "-lamdhip64"], check=True)
import torch  # noqa: E402
from torecsys_amd import functional as F_, _abi  # noqa: E402
_abi.LIB_PATH = out          # before the first call: _abi.load() opens it lazily
dev = torch.device("cuda:0")
Bn, N, E, L = 65536, 39, 64, 6
g = torch.Generator().manual_seed(0)
x = torch.randn(Bn, N, E, generator=g).bfloat16().to(dev).requires_grad_()
W = (torch.randn(L, E, E, generator=g) / 8).bfloat16().to(dev).requires_grad_()
b = torch.zeros(L, E).bfloat16().to(dev).requires_grad_()
for _ in range(2):
    y = F_.cross_network(x, W, b)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
y = F_.cross_network(x, W, b)
gy = torch.ones_like(y)
ev[0].record()
y.backward(gy)
ev[1].record()
torch.cuda.synchronize()
print("backward (all kernels, traced build)", EXTRA, "%.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
lib = ctypes.CDLL(out)
buf = (ctypes.c_longlong * 2048)()
assert lib.trs_debug_b3_trace(buf) == 0
t = list(buf)
ch, dw = t[:1024], t[1024:]
print("wall_clock64 ticks (100 MHz: 10 ns each); chain wave 0: [start, afterA, (before,after barrier) x3]; dW wave: (arrive,release) x4")
for gi in (3, 4, 50, 51):
    c = ch[gi * 8: gi * 8 + 8]
    d = dw[gi * 8: gi * 8 + 8]
    base = c[0]
    print("group", gi, "chain", [v - base for v in c], "dW", [v - base for v in d], "next group start", ch[(gi + 1) * 8] - base)
