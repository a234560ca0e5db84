"""Golden vectors for the pair-pattern layers of SURVEY.md 8f N3 (OuterProductNetwork, AttentionalFactorizationMachine,
BilinearInteraction), captured from the REAL reference in this container (same recipe as make_golden.py).
Run:  python tests/golden/make_golden_pairs.py    (needs /root/reference; writes tests/golden/pairs.npz)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, npy, save  # noqa: E402

PAIR_SHAPES = [(8, 4, 128), (16, 6, 64), (32, 12, 8), (32, 10, 16), (8, 12, 64), (4, 39, 64)]
HEAVY_MAX = 300_000      # per-pair E x E parameters ('mat' kernel, 'each' weights) are skipped above this many elements


def heavy_ok(N, E):
    return N * (N - 1) // 2 * E * E <= HEAVY_MAX



def gen(layers_mod, out):
    for (B, N, E) in PAIR_SHAPES:
        g = torch.Generator().manual_seed(7000 + B * 3 + N * 13 + E)
        tag = f"{B}_{N}_{E}"
        P = N * (N - 1) // 2
        x = torch.randn(B, N, E, generator=g)
        out[f"x/{tag}"] = npy(x)
        # N3a OuterProductNetworkLayer (outer_product_network.py:36-129), three kernel types
        for kt in ("mat", "vec", "num"):
            if kt == "mat" and not heavy_ok(N, E):
                continue
            torch.manual_seed(7100 + B + N + E)
            lay = layers_mod.OuterProductNetworkLayer(embed_size=E, num_fields=N, kernel_type=kt)
            xa = x.clone().requires_grad_()
            xa_n = xa.refine_names('B', 'N', 'E')
            y = lay(xa_n)
            go = torch.randn(B, P, generator=g)
            (y.rename(None) * go).sum().backward()
            out[f"opn_{kt}/{tag}/kernel"] = npy(lay.kernel)
            out[f"opn_{kt}/{tag}/out"] = npy(y)
            out[f"opn_{kt}/{tag}/names"] = np.array(list(y.names))
            out[f"opn_{kt}/{tag}/gout"] = npy(go)
            out[f"opn_{kt}/{tag}/gx"] = npy(xa.grad)
            out[f"opn_{kt}/{tag}/gkernel"] = npy(lay.kernel.grad)
        # N3b AttentionalFactorizationMachineLayer (attentional_factorization_machine.py:49-125), dropout 0
        A = 16 if E >= 16 else 8
        torch.manual_seed(7200 + B + N + E)
        lay = layers_mod.AttentionalFactorizationMachineLayer(embed_size=E, num_fields=N, attn_size=A, dropout_p=0.0)
        xa = (0.5 * x).clone().requires_grad_()
        y, attn = lay(xa.refine_names('B', 'N', 'E'))
        go = torch.randn(B, E, generator=g)
        ga = torch.randn(B, P, 1, generator=g) * 0.1
        ((y.rename(None) * go).sum() + (attn.rename(None) * ga).sum()).backward()
        out[f"afm/{tag}/x"] = npy(0.5 * x)
        out[f"afm/{tag}/W1"] = npy(lay.attention.Linear.weight)
        out[f"afm/{tag}/b1"] = npy(lay.attention.Linear.bias)
        out[f"afm/{tag}/W2"] = npy(lay.attention.OutProj.weight)
        out[f"afm/{tag}/b2"] = npy(lay.attention.OutProj.bias)
        out[f"afm/{tag}/out"] = npy(y)
        out[f"afm/{tag}/attn"] = npy(attn)
        out[f"afm/{tag}/names"] = np.array([str(n) for n in y.names])
        out[f"afm/{tag}/attn_names"] = np.array([str(n) for n in attn.names])
        out[f"afm/{tag}/gout"] = npy(go)
        out[f"afm/{tag}/gattn"] = npy(ga)
        out[f"afm/{tag}/gx"] = npy(xa.grad)
        out[f"afm/{tag}/gW1"] = npy(lay.attention.Linear.weight.grad)
        out[f"afm/{tag}/gb1"] = npy(lay.attention.Linear.bias.grad)
        out[f"afm/{tag}/gW2"] = npy(lay.attention.OutProj.weight.grad)
        out[f"afm/{tag}/gb2"] = npy(lay.attention.OutProj.bias.grad)
        # N3c BilinearInteractionLayer (bilinear_interaction.py:179-255), 'all' and 'each'
        for bt in ("all", "each"):
            if bt == "each" and not heavy_ok(N, E):
                continue
            torch.manual_seed(7300 + B + N + E)
            lay = layers_mod.BilinearInteractionLayer(embed_size=E, num_fields=N, bilinear_type=bt, bias=True)
            xa = x.clone().requires_grad_()
            y = lay(xa.refine_names('B', 'N', 'E'))
            go = torch.randn(B, P, E, generator=g)
            (y.rename(None) * go).sum().backward()
            out[f"bil_{bt}/{tag}/W"] = npy(lay.bilinear.weight)
            out[f"bil_{bt}/{tag}/b"] = npy(lay.bilinear.bias)
            out[f"bil_{bt}/{tag}/out"] = npy(y)
            out[f"bil_{bt}/{tag}/names"] = np.array(list(y.names))
            out[f"bil_{bt}/{tag}/gout"] = npy(go)
            out[f"bil_{bt}/{tag}/gx"] = npy(xa.grad)
            out[f"bil_{bt}/{tag}/gW"] = npy(lay.bilinear.weight.grad)
            out[f"bil_{bt}/{tag}/gb"] = npy(lay.bilinear.bias.grad)
    # reference quirks worth pinning
    for name, ctor in (("bil_nobias", lambda: layers_mod.BilinearInteractionLayer(8, 4, "all", bias=False)),
                       ("bil_interaction", lambda: layers_mod.BilinearInteractionLayer(8, 4, "interaction")),
                       ("opn_badtype", lambda: layers_mod.OuterProductNetworkLayer(8, 4, "cube"))):
        try:
            ctor()
            raised = ""
        except Exception as e:  # noqa: BLE001
            raised = type(e).__name__
        out[f"raises/{name}"] = np.array([raised])


def main():
    _, layers_mod, _ = import_reference()
    d = {}
    gen(layers_mod, d)
    save("pairs.npz", d)
    for k in sorted(d):
        if k.startswith("raises/") or k.endswith("names"):
            print(k, d[k])


if __name__ == "__main__":
    main()
