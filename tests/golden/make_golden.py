#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only at /root/reference).

Runs ONLY in the build container (the reference never travels to the GPU box).
Imports the reference's modules with the stub recipe of SURVEY.md §8c (skip
``torecsys/__init__.py`` -- it pulls pytorch_lightning -- and stub torchvision),
runs each hot-path symbol forward and backward on CPU fp32 with fixed seeds and
dropout 0, and stores inputs + parameters + outputs + gradients as arrays:

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures hold data only (arrays); no reference source is copied.
"""
import importlib
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    pkg = types.ModuleType("torecsys")
    pkg.__path__ = [os.path.join(REF, "torecsys")]
    sys.modules["torecsys"] = pkg
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    inputs = importlib.import_module("torecsys.inputs")
    layers = importlib.import_module("torecsys.layers")
    models = importlib.import_module("torecsys.models")
    return inputs, layers, models


def npy(t):
    return t.detach().rename(None).cpu().numpy().copy()


def rand_field_sizes(g, n, lo=3, hi=11):
    return [int(v) for v in torch.randint(lo, hi + 1, (n,), generator=g)]


def rand_idx(g, B, field_sizes):
    cols = [torch.randint(0, fs, (B, 1), generator=g) for fs in field_sizes]
    return torch.cat(cols, dim=1)


LAYER_SHAPES = [(8, 4, 128), (16, 6, 64), (32, 12, 8), (32, 10, 16), (8, 39, 64)]


def gen_inputs(inputs_mod, out):
    for (B, N, E) in LAYER_SHAPES:
        g = torch.Generator().manual_seed(1000 + B * 7 + N * 3 + E)
        torch.manual_seed(2000 + B + N + E)
        fs = rand_field_sizes(g, N)
        idx = rand_idx(g, B, fs)
        tag = f"{B}_{N}_{E}"
        # I2 MultiIndicesEmbedding (multi_indices_emb.py)
        m = inputs_mod.MultiIndicesEmbedding(embed_size=E, field_sizes=fs)
        gout = torch.randn(B, N, E, generator=g)
        y = m(idx)
        (y.rename(None) * gout).sum().backward()
        out[f"multi/{tag}/field_sizes"] = np.array(fs, dtype=np.int64)
        out[f"multi/{tag}/idx"] = npy(idx)
        out[f"multi/{tag}/weight"] = npy(m.embedding.weight)
        out[f"multi/{tag}/out"] = npy(y)
        out[f"multi/{tag}/names"] = np.array(list(y.names))
        out[f"multi/{tag}/gout"] = npy(gout)
        out[f"multi/{tag}/gweight"] = npy(m.embedding.weight.grad)
        out[f"multi/{tag}/offsets"] = npy(m.offsets).reshape(-1)
        mf = inputs_mod.MultiIndicesEmbedding(embed_size=E, field_sizes=fs, flatten=True)
        yf = mf(idx)
        out[f"multi/{tag}/flatten_shape"] = np.array(list(yf.shape), dtype=np.int64)
        out[f"multi/{tag}/length"] = np.array([len(m), len(mf)], dtype=np.int64)
        # first-order table, E=1 (tests/test_trainer.py:55 style)
        m1 = inputs_mod.MultiIndicesEmbedding(embed_size=1, field_sizes=fs)
        y1 = m1(idx)
        out[f"multi1/{tag}/weight"] = npy(m1.embedding.weight)
        out[f"multi1/{tag}/out"] = npy(y1)
        # I1 SingleIndexEmbedding (single_index_emb.py), int32 indices, with padding_idx
        s = inputs_mod.SingleIndexEmbedding(embed_size=E, field_size=fs[0] + 2, padding_idx=0)
        sidx = torch.randint(0, fs[0] + 2, (B, 1), generator=g).to(torch.int32)
        ys = s(sidx)
        gs = torch.randn(B, 1, E, generator=g)
        (ys.rename(None) * gs).sum().backward()
        out[f"single/{tag}/idx"] = npy(sidx)
        out[f"single/{tag}/weight"] = npy(s.embedding.weight)
        out[f"single/{tag}/out"] = npy(ys)
        out[f"single/{tag}/gout"] = npy(gs)
        out[f"single/{tag}/gweight"] = npy(s.embedding.weight.grad)
        out[f"single/{tag}/names"] = np.array(list(ys.names))
        # I3 MultiIndicesFieldAwareEmbedding (multi_indices_field_aware_emb.py)
        Ef = E if N <= 12 else 16
        fa = inputs_mod.MultiIndicesFieldAwareEmbedding(embed_size=Ef, field_sizes=fs)
        yfa = fa(idx)
        gfa = torch.randn(B, N * N, Ef, generator=g)
        (yfa.rename(None) * gfa).sum().backward()
        out[f"fa/{tag}/weights"] = np.stack([npy(e.weight) for e in fa.embeddings])
        out[f"fa/{tag}/out_checksum"] = np.array([float(npy(yfa).astype(np.float64).sum())])
        # store a strided subset of rows (full (B,N*N,E) is big for N=39)
        out[f"fa/{tag}/out_sub"] = npy(yfa)[:, :: max(1, N // 3)]
        out[f"fa/{tag}/out_sub_stride"] = np.array([max(1, N // 3)], dtype=np.int64)
        out[f"fa/{tag}/names"] = np.array(list(yfa.names))
        out[f"fa/{tag}/gout_seed"] = np.array([0], dtype=np.int64)
        # gradient of table 1 under gout = ones (compact to store)
        for e in fa.embeddings:
            e.weight.grad = None
        fa(idx).rename(None).sum().backward()
        out[f"fa/{tag}/gweight1_ones"] = npy(fa.embeddings[1].weight.grad)


def gen_layers(layers_mod, out):
    for (B, N, E) in LAYER_SHAPES:
        g = torch.Generator().manual_seed(3000 + B * 5 + N * 11 + E)
        tag = f"{B}_{N}_{E}"
        x = torch.randn(B, N, E, generator=g)
        # F1 FM (factorization_machine.py)
        xa = x.clone().requires_grad_()
        lay = layers_mod.FactorizationMachineLayer(dropout_p=0.0)
        y = lay(xa)
        go = torch.randn(B, E, generator=g)
        (y.rename(None) * go).sum().backward()
        out[f"fm/{tag}/x"] = npy(x)
        out[f"fm/{tag}/out"] = npy(y)
        out[f"fm/{tag}/names"] = np.array(list(y.names))
        out[f"fm/{tag}/gout"] = npy(go)
        out[f"fm/{tag}/gx"] = npy(xa.grad)
        # F5 IPN (inner_product_network.py)
        xa = x.clone().requires_grad_()
        lay = layers_mod.InnerProductNetworkLayer(num_fields=N)
        y = lay(xa)
        go = torch.randn(B, N * (N - 1) // 2, generator=g)
        (y.rename(None) * go).sum().backward()
        out[f"ipn/{tag}/out"] = npy(y)
        out[f"ipn/{tag}/names"] = np.array(list(y.names))
        out[f"ipn/{tag}/gout"] = npy(go)
        out[f"ipn/{tag}/gx"] = npy(xa.grad)
        # F2 FFM (field_aware_factorization_machine.py); input is (B, N*N, E)
        Ef = E if N <= 12 else 16
        Bf = B if N <= 12 else 4
        xf = torch.randn(Bf, N * N, Ef, generator=g)
        xa = xf.clone().requires_grad_()
        lay = layers_mod.FieldAwareFactorizationMachineLayer(num_fields=N, dropout_p=0.0)
        y = lay(xa)
        go = torch.randn(Bf, N * (N - 1) // 2, Ef, generator=g)
        (y.rename(None) * go).sum().backward()
        out[f"ffm/{tag}/x"] = npy(xf)
        out[f"ffm/{tag}/out"] = npy(y)
        out[f"ffm/{tag}/names"] = np.array(list(y.names))
        out[f"ffm/{tag}/gout"] = npy(go)
        out[f"ffm/{tag}/gx"] = npy(xa.grad)
        # F3 Cross (cross_network.py); 4 layers as tests/test_layers.py:188-212, 6 for the 39-field shape
        L = 6 if N == 39 else 4
        torch.manual_seed(4000 + B + N + E)
        lay = layers_mod.CrossNetworkLayer(inputs_size=E, num_layers=L)
        xa = (0.5 * x).clone().requires_grad_()
        y = lay(xa)
        go = torch.randn(B, N, E, generator=g)
        (y.rename(None) * go).sum().backward()
        out[f"cross/{tag}/x"] = npy(0.5 * x)
        out[f"cross/{tag}/W"] = np.stack([npy(l.weight) for l in lay.model])
        out[f"cross/{tag}/b"] = np.stack([npy(l.bias) for l in lay.model])
        out[f"cross/{tag}/out"] = npy(y)
        out[f"cross/{tag}/names"] = np.array(list(y.names))
        out[f"cross/{tag}/gout"] = npy(go)
        out[f"cross/{tag}/gx"] = npy(xa.grad)
        out[f"cross/{tag}/gW"] = np.stack([npy(l.weight.grad) for l in lay.model])
        out[f"cross/{tag}/gb"] = np.stack([npy(l.bias.grad) for l in lay.model])
    # cross on a 2-D input: the reference's ('B','O') branch (cross_network.py:82-83) is
    # unreachable -- einsum('ijk,ijk->ijk') at :78 raises RuntimeError for 2-D inputs.
    lay = layers_mod.CrossNetworkLayer(inputs_size=16, num_layers=3)
    try:
        lay(torch.randn(8, 16))
        raised = ""
    except Exception as e:  # noqa: BLE001
        raised = type(e).__name__
    out["cross2d/raises"] = np.array([raised])


def gen_cin(layers_mod, out):
    # tests/test_layers.py:155-185 uses layer_sizes=[32,64,32], BN on; we add eval / direct / no-BN variants
    cases = [
        ("a", 8, 4, 128, [32, 64, 32], False, True, True),
        ("b", 16, 6, 64, [32, 64, 32], False, True, True),
        ("c", 32, 12, 8, [32, 64, 32], False, True, True),
        ("d", 16, 6, 64, [16, 8], True, True, True),       # is_direct
        ("e", 16, 6, 64, [16, 8], False, False, False),    # no bias, no BN
        ("f", 8, 39, 64, [16, 16, 16], False, True, True),
    ]
    for (name, B, N, E, sizes, direct, use_bias, use_bn) in cases:
        g = torch.Generator().manual_seed(5000 + B + N + E + len(sizes))
        torch.manual_seed(6000 + B + N + E)
        lay = layers_mod.CompressInteractionNetworkLayer(
            embed_size=E, num_fields=N, output_size=3, layer_sizes=list(sizes),
            is_direct=direct, use_bias=use_bias, use_batchnorm=use_bn, activation=nn.ReLU())
        if use_bn:  # non-trivial affine + running stats
            for seq in lay.model:
                seq.Batchnorm.weight.data.uniform_(0.5, 1.5, generator=g)
                seq.Batchnorm.bias.data.normal_(0, 0.1, generator=g)
        x = 0.7 * torch.randn(B, N, E, generator=g)
        pre = f"cin/{name}"
        out[f"{pre}/cfg"] = np.array([B, N, E, int(direct), int(use_bias), int(use_bn)], dtype=np.int64)
        out[f"{pre}/layer_sizes"] = np.array(sizes, dtype=np.int64)
        out[f"{pre}/x"] = npy(x)
        for i, seq in enumerate(lay.model):
            out[f"{pre}/conv_w{i}"] = npy(seq.Conv1d.weight)
            if use_bias:
                out[f"{pre}/conv_b{i}"] = npy(seq.Conv1d.bias)
            if use_bn:
                out[f"{pre}/bn_w{i}"] = npy(seq.Batchnorm.weight)
                out[f"{pre}/bn_b{i}"] = npy(seq.Batchnorm.bias)
        out[f"{pre}/fc_w"] = npy(lay.fc.weight)
        out[f"{pre}/fc_b"] = npy(lay.fc.bias)
        # training-mode forward/backward
        lay.train()
        xa = x.clone().requires_grad_()
        y = lay(xa)
        go = torch.randn(B, 3, generator=g)
        (y.rename(None) * go).sum().backward()
        out[f"{pre}/train_out"] = npy(y)
        out[f"{pre}/names"] = np.array(list(y.names))
        out[f"{pre}/gout"] = npy(go)
        out[f"{pre}/train_gx"] = npy(xa.grad)
        for i, seq in enumerate(lay.model):
            out[f"{pre}/train_gconv_w{i}"] = npy(seq.Conv1d.weight.grad)
            if use_bias:
                out[f"{pre}/train_gconv_b{i}"] = npy(seq.Conv1d.bias.grad)
            if use_bn:
                out[f"{pre}/train_gbn_w{i}"] = npy(seq.Batchnorm.weight.grad)
                out[f"{pre}/train_gbn_b{i}"] = npy(seq.Batchnorm.bias.grad)
                out[f"{pre}/run_mean{i}"] = npy(seq.Batchnorm.running_mean)
                out[f"{pre}/run_var{i}"] = npy(seq.Batchnorm.running_var)
        out[f"{pre}/train_gfc_w"] = npy(lay.fc.weight.grad)
        # eval-mode forward (uses the running stats just updated)
        lay.eval()
        xa = x.clone().requires_grad_()
        ye = lay(xa)
        (ye.rename(None) * go).sum().backward()
        out[f"{pre}/eval_out"] = npy(ye)
        out[f"{pre}/eval_gx"] = npy(xa.grad)


def gen_models(inputs_mod, models_mod, out):
    def mlp_params(dnn):
        ws, bs = [], []
        for mod in dnn.model:
            if isinstance(mod, nn.Linear):
                ws.append(npy(mod.weight))
                bs.append(npy(mod.bias))
        return ws, bs

    def mlp_grads(dnn):
        return [npy(mod.weight.grad) for mod in dnn.model if isinstance(mod, nn.Linear)]

    for (B, N, E) in [(32, 10, 16), (16, 39, 64), (16, 6, 64)]:
        g = torch.Generator().manual_seed(7000 + B + N + E)
        torch.manual_seed(8000 + B + N + E)
        fs = rand_field_sizes(g, N)
        idx = rand_idx(g, B, fs)
        tag = f"{B}_{N}_{E}"
        emb = inputs_mod.MultiIndicesEmbedding(embed_size=E, field_sizes=fs)
        emb.embedding.weight.data.mul_(0.3)
        feat = inputs_mod.MultiIndicesEmbedding(embed_size=1, field_sizes=fs)
        out[f"model/{tag}/field_sizes"] = np.array(fs, dtype=np.int64)
        out[f"model/{tag}/idx"] = npy(idx)
        out[f"model/{tag}/emb_w"] = npy(emb.embedding.weight)
        out[f"model/{tag}/feat_w"] = npy(feat.embedding.weight)
        gout = torch.randn(B, 1, generator=g)
        out[f"model/{tag}/gout"] = npy(gout)

        def run(model, two_inputs, key):
            emb.zero_grad()
            feat.zero_grad()
            model.zero_grad()
            if two_inputs:
                y = model(feat(idx), emb(idx))
            else:
                y = model(emb(idx))
            (y * gout).sum().backward()
            out[f"model/{tag}/{key}_out"] = npy(y)
            out[f"model/{tag}/{key}_gemb"] = npy(emb.embedding.weight.grad)
            if two_inputs:
                out[f"model/{tag}/{key}_gfeat"] = npy(feat.embedding.weight.grad)

        # M1 FactorizationMachineModel (models/ctr/factorization_machine.py)
        m1 = models_mod.FactorizationMachineModel(use_bias=True, dropout_p=0.0)
        out[f"model/{tag}/fm_bias"] = npy(m1.bias)
        run(m1, True, "fm")
        # M2 DeepFM (models/ctr/deep_fm.py)
        m2 = models_mod.DeepFactorizationMachineModel(
            embed_size=E, num_fields=N, deep_layer_sizes=[32, 16], fm_dropout_p=0.0)
        ws, bs = mlp_params(m2.deep)
        for i, (w, b) in enumerate(zip(ws, bs)):
            out[f"model/{tag}/deepfm_w{i}"] = w
            out[f"model/{tag}/deepfm_b{i}"] = b
        run(m2, True, "deepfm")
        out[f"model/{tag}/deepfm_gw0"] = mlp_grads(m2.deep)[0]
        # M3 DCN (models/ctr/deep_and_cross_network.py); inputs_size = E (tests/test_models.py:57-59)
        m3 = models_mod.DeepAndCrossNetworkModel(
            inputs_size=E, num_fields=N, deep_output_size=8, deep_layer_sizes=[16, 16],
            cross_num_layers=3, output_size=1)
        ws, bs = mlp_params(m3.deep)
        for i, (w, b) in enumerate(zip(ws, bs)):
            out[f"model/{tag}/dcn_w{i}"] = w
            out[f"model/{tag}/dcn_b{i}"] = b
        out[f"model/{tag}/dcn_cross_W"] = np.stack([npy(l.weight) for l in m3.cross.model])
        out[f"model/{tag}/dcn_cross_b"] = np.stack([npy(l.bias) for l in m3.cross.model])
        out[f"model/{tag}/dcn_fc_w"] = npy(m3.fc.weight)
        out[f"model/{tag}/dcn_fc_b"] = npy(m3.fc.bias)
        run(m3, False, "dcn")
        out[f"model/{tag}/dcn_gcross_W"] = np.stack([npy(l.weight.grad) for l in m3.cross.model])
        # M4 xDeepFM (models/ctr/xdeep_fm.py), train-mode BN
        m4 = models_mod.XDeepFactorizationMachineModel(
            embed_size=E, num_fields=N, cin_layer_sizes=[8, 8], deep_layer_sizes=[16, 8])
        m4.train()
        ws, bs = mlp_params(m4.deep)
        for i, (w, b) in enumerate(zip(ws, bs)):
            out[f"model/{tag}/xdfm_w{i}"] = w
            out[f"model/{tag}/xdfm_b{i}"] = b
        for i, seq in enumerate(m4.cin.model):
            out[f"model/{tag}/xdfm_conv_w{i}"] = npy(seq.Conv1d.weight)
            out[f"model/{tag}/xdfm_conv_b{i}"] = npy(seq.Conv1d.bias)
            out[f"model/{tag}/xdfm_bn_w{i}"] = npy(seq.Batchnorm.weight)
            out[f"model/{tag}/xdfm_bn_b{i}"] = npy(seq.Batchnorm.bias)
        out[f"model/{tag}/xdfm_fc_w"] = npy(m4.cin.fc.weight)
        out[f"model/{tag}/xdfm_fc_b"] = npy(m4.cin.fc.bias)
        out[f"model/{tag}/xdfm_bias"] = npy(m4.bias)
        run(m4, True, "xdfm")


def gen_state_dict_keys(inputs_mod, layers_mod, out):
    """Parameter / buffer names a drop-in must keep (SURVEY §5 checkpoint row)."""
    keys = {
        "multi": list(inputs_mod.MultiIndicesEmbedding(embed_size=4, field_sizes=[3, 4]).state_dict().keys()),
        "single": list(inputs_mod.SingleIndexEmbedding(embed_size=4, field_size=5).state_dict().keys()),
        "fa": list(inputs_mod.MultiIndicesFieldAwareEmbedding(embed_size=4, field_sizes=[3, 4]).state_dict().keys()),
        "fm": list(layers_mod.FactorizationMachineLayer().state_dict().keys()),
        "ffm": list(layers_mod.FieldAwareFactorizationMachineLayer(num_fields=3).state_dict().keys()),
        "ipn": list(layers_mod.InnerProductNetworkLayer(num_fields=3).state_dict().keys()),
        "cross": list(layers_mod.CrossNetworkLayer(inputs_size=4, num_layers=2).state_dict().keys()),
        "cin": list(layers_mod.CompressInteractionNetworkLayer(
            embed_size=4, num_fields=3, output_size=1, layer_sizes=[2, 2]).state_dict().keys()),
    }
    for k, v in keys.items():
        out[f"keys/{k}"] = np.array(v if v else [""], dtype=str)


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print(f"{name}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    inputs_mod, layers_mod, models_mod = import_reference()
    d = {}
    gen_inputs(inputs_mod, d)
    save("inputs.npz", d)
    d = {}
    gen_layers(layers_mod, d)
    save("layers.npz", d)
    d = {}
    gen_cin(layers_mod, d)
    save("cin.npz", d)
    d = {}
    gen_models(inputs_mod, models_mod, d)
    save("models.npz", d)
    d = {}
    gen_state_dict_keys(inputs_mod, layers_mod, d)
    save("keys.npz", d)


if __name__ == "__main__":
    main()
