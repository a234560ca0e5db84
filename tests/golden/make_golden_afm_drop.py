"""Golden vectors for AttentionalFactorizationMachineLayer at its DEFAULT configuration -- training mode with dropout on the
attention scores and on the output (attentional_factorization_machine.py:53, 82, 84, 105-120) -- captured from the REAL
reference in this container (recipe of make_golden.py).  The two dropout masks the reference drew are recorded by forward
hooks on its nn.Dropout modules, so the oracle / the HIP kernel can be given exactly the same masks.
Run:  python tests/golden/make_golden_afm_drop.py    (needs /root/reference; writes tests/golden/afm_drop.npz)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, npy, save  # noqa: E402

SHAPES = [(8, 4, 128, 0.25), (16, 6, 64, 0.1), (32, 12, 8, 0.5), (8, 12, 64, 0.1), (4, 39, 64, 0.1)]


def gen(layers_mod, out):
    for (B, N, E, p) in SHAPES:
        g = torch.Generator().manual_seed(9000 + B * 3 + N * 13 + E)
        tag = f"{B}_{N}_{E}"
        P = N * (N - 1) // 2
        A = 16 if E >= 16 else 8
        torch.manual_seed(9200 + B + N + E)
        lay = layers_mod.AttentionalFactorizationMachineLayer(embed_size=E, num_fields=N, attn_size=A, dropout_p=p)
        lay.train()
        seen = {}

        def hook(name):
            def f(mod, inp, res):
                seen[name] = (inp[0].detach().rename(None).clone(), res.detach().rename(None).clone())
            return f
        h1 = lay.attention.Dropout.register_forward_hook(hook("score"))
        h2 = lay.dropout.register_forward_hook(hook("out"))
        x = 0.5 * torch.randn(B, N, E, generator=g)
        xa = x.clone().requires_grad_()
        y, attn = lay(xa.refine_names('B', 'N', 'E'))
        go = torch.randn(B, E, generator=g)
        ga = torch.randn(B, P, 1, generator=g) * 0.1
        ((y.rename(None) * go).sum() + (attn.rename(None) * ga).sum()).backward()
        h1.remove()
        h2.remove()
        s_in, s_out = seen["score"]          # softmax scores (all > 0) and the dropped scores
        o_in, o_out = seen["out"]
        out[f"{tag}/p"] = np.array([p], dtype=np.float64)
        out[f"{tag}/x"] = npy(x)
        out[f"{tag}/W1"] = npy(lay.attention.Linear.weight)
        out[f"{tag}/b1"] = npy(lay.attention.Linear.bias)
        out[f"{tag}/W2"] = npy(lay.attention.OutProj.weight)
        out[f"{tag}/b2"] = npy(lay.attention.OutProj.bias)
        out[f"{tag}/score_keep"] = (s_out != 0).squeeze(-1).numpy().astype(np.uint8)       # (B, P)
        out[f"{tag}/out_keep"] = ((o_out != 0) | (o_in == 0)).numpy().astype(np.uint8)     # (B, E)
        out[f"{tag}/out_before_dropout"] = npy(o_in)
        out[f"{tag}/out"] = npy(y)
        out[f"{tag}/attn"] = npy(attn)
        out[f"{tag}/gout"] = npy(go)
        out[f"{tag}/gattn"] = npy(ga)
        out[f"{tag}/gx"] = npy(xa.grad)
        out[f"{tag}/gW1"] = npy(lay.attention.Linear.weight.grad)
        out[f"{tag}/gb1"] = npy(lay.attention.Linear.bias.grad)
        out[f"{tag}/gW2"] = npy(lay.attention.OutProj.weight.grad)
        out[f"{tag}/gb2"] = npy(lay.attention.OutProj.bias.grad)


def main():
    _, layers_mod, _ = import_reference()
    d = {}
    gen(layers_mod, d)
    save("afm_drop.npz", d)
    for k in sorted(d):
        if k.endswith("/p"):
            kk = k[:-2]
            print(kk, "p", d[k][0], "kept", float(d[kk + "/score_keep"].mean()))


if __name__ == "__main__":
    main()
