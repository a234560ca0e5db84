#!/usr/bin/env python3
"""Build-container only: time the REAL reference (p768lwy3/torecsys imported from /root/reference with the stub recipe
of SURVEY.md section 8c) against the oracle restatement (oracle/cpu_ref.py) on identical DeepFM inputs, fwd+bwd, fp32
CPU.  The ratio shows how faithful the travelling CPU baseline (bench.py cpu_baseline, kind "port") is to the
reference's own CPU path.  Never runs on the GPU box (the reference does not travel).

    python tools/ref_vs_oracle.py [--batch 16384] [--threads 8]
"""
import argparse
import importlib
import os
import sys
import time
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as O  # noqa: E402


def import_reference():
    pkg = types.ModuleType("torecsys")
    pkg.__path__ = ["/root/reference/torecsys"]
    sys.modules["torecsys"] = pkg
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    return importlib.import_module("torecsys.inputs"), importlib.import_module("torecsys.models")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    inputs_mod, models_mod = import_reference()
    N, E, V = 39, 64, 1_000_000
    per = V // N
    sizes = [per] * (N - 1) + [V - per * (N - 1)]
    g = torch.Generator().manual_seed(4321)
    idx = torch.cat([torch.randint(0, f, (a.batch, 1), generator=g) for f in sizes], 1)
    y = (torch.rand(a.batch, 1, generator=g) < 0.25).float()
    torch.manual_seed(0)
    emb = inputs_mod.MultiIndicesEmbedding(embed_size=E, field_sizes=sizes)
    feat = inputs_mod.MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
    model = models_mod.DeepFactorizationMachineModel(embed_size=E, num_fields=N, deep_layer_sizes=[400, 400, 400],
                                                     fm_dropout_p=0.0, deep_dropout_p=[0.0, 0.0, 0.0])
    params = list(emb.parameters()) + list(feat.parameters()) + list(model.parameters())

    def ref_step():
        for p in params:
            p.grad = None
        out = model(feat_inputs=feat(idx), emb_inputs=emb(idx))
        nn.functional.binary_cross_entropy_with_logits(out, y).backward()

    w = emb.embedding.weight.detach().clone().requires_grad_()
    w1 = feat.embedding.weight.detach().clone().requires_grad_()
    lin = [m for m in model.deep.model if isinstance(m, nn.Linear)]
    ws = [m.weight.detach().clone().requires_grad_() for m in lin]
    bs = [m.bias.detach().clone().requires_grad_() for m in lin]
    off = O.field_offsets(sizes)

    def oracle_step():
        for t in [w, w1, *ws, *bs]:
            t.grad = None
        logit = O.deepfm_model(O.multi_indices_embedding(w1, idx, off), O.multi_indices_embedding(w, idx, off), ws, bs)
        nn.functional.binary_cross_entropy_with_logits(logit, y).backward()

    def timeit(fn):
        fn()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        return (time.perf_counter() - t0) / a.steps

    import warnings
    warnings.simplefilter("ignore")
    tr, to = timeit(ref_step), timeit(oracle_step)
    ref_step(); oracle_step()
    same = float((emb.embedding.weight.grad - w.grad).abs().max() / w.grad.abs().max())
    print(f"threads {a.threads}  batch {a.batch}: reference {tr*1e3:.1f} ms/step ({a.batch/tr:.0f} samples/s), "
          f"oracle {to*1e3:.1f} ms/step ({a.batch/to:.0f} samples/s), reference/oracle time = {tr/to:.3f}; "
          f"embedding-gradient max rel diff {same:.2e}")


if __name__ == "__main__":
    main()
