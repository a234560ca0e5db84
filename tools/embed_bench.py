"""Stand-alone timing of the fused lookup + FM launch (trs_embed_fm) at the BASELINE shape on a table of --rows rows:
median of HIP-event timings on the launch stream, with and without the (B,N,E) block written.

    python tools/embed_bench.py [--rows 32000000] [--batch 65536] [--zipf] [--layout uniform|skewed]
    TRS_EMBED_PIPE=4|8: the pipelined walk (ids of the next chunk requested beside the current chunk's rows)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torecsys_amd import _abi  # noqa: E402
from torecsys_amd import functional as F_  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=32_000_000)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--zipf", action="store_true")
ap.add_argument("--layout", default="uniform")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N, E = a.batch, 39, 64
sizes = bench.field_sizes(a.rows, N, a.layout)
gen = torch.Generator().manual_seed(99)
idx = bench.synth_indices(B, sizes, gen, a.zipf).to(dev)
off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.tensor(sizes), 0)[:-1]]).to(dev)
w = torch.empty(a.rows, E, dtype=torch.bfloat16, device=dev).normal_()
for fm_only in (False, True):
    fn = lambda: F_._EmbedFM.apply(w, idx, off, None, not fm_only)  # noqa: E731
    for _ in range(3):
        fn()
    _abi.time_kernel("trs_embed_fm", True, expect=22, every=1)
    for _ in range(20):
        fn()
    ts = sorted(_abi.kernel_times_ms("trs_embed_fm"))
    _abi.time_kernel("trs_embed_fm", False)
    med = ts[len(ts) // 2] * 1e-3
    alg = B * N * (8 + E * 2) + (0 if fm_only else B * N * E * 2) + B * E * 2
    print(f"rows {a.rows:>9d} pipe {os.environ.get('TRS_EMBED_PIPE', '0')} {'fm-only' if fm_only else 'with block'}: "
          f"median {med * 1e6:7.2f} us  min {ts[0] * 1e3:7.2f}  {alg / med / 1e9:7.1f} GB/s  frac {alg / med / 8e12:.4f}")
