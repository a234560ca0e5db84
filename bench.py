#!/usr/bin/env python3
"""bench.py -- CTR samples/s, forward + backward, DeepFM (39 Criteo-shaped fields x dim 64) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: index lookup (E=64 table + E=1
first-order table) fused with the FM second-order term, the DeepFM MLP [400,400,400] (nn.Linear ->
hipBLASLt, outside the hand-written path but inside the timed step), BCE-with-logits loss, and the full
backward including the dense embedding-table gradients.  No optimizer step: the metric is fwd+bwd.
Inputs (indices, labels) are resident in HBM before the timed region.

N = 1: BASELINE.json configs[1] (V = 1 M rows, B = 65 536, bf16).
N > 1: configs[4] scaled weakly: 125 M rows and 65 536 samples per GPU, table row-sharded over the ranks,
       lookup by RCCL all-to-all, gradient returned by the reverse all-to-all (torecsys_amd/dist.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the fused lookup+FM
forward kernel, timed live with HIP events on its launch stream) and `cpu_baseline` (the CPU oracle timed
on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if "--no-tunableop" not in sys.argv:
    # The MLP GEMMs (nn.Linear, outside the hand-written path) go through PyTorch's TunableOp: solutions tuned
    # once on an MI355X for exactly these shapes are shipped as ONE file (torecsys_amd/tuning/tunableop_mi355x.csv);
    # TunableOp reads / rewrites "<name><device ordinal>.csv", so every rank works on its own scratch copy.  Shapes
    # missing from the file (or a library-version mismatch) are tuned during warm-up (~15 s).
    import shutil
    import tempfile
    _rank = os.environ.get("LOCAL_RANK", "0")
    _tdir = os.path.join(tempfile.gettempdir(), "trs_tunableop_%d" % os.getuid())
    os.makedirs(_tdir, exist_ok=True)
    try:
        shutil.copyfile(os.path.join(ROOT, "torecsys_amd", "tuning", "tunableop_mi355x.csv"),
                        os.path.join(_tdir, "tunableop_results%s.csv" % _rank))
    except OSError:
        pass
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "60")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "30")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(_tdir, "tunableop_results.csv"))

import torch
import torch.nn as nn

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU")
    ap.add_argument("--fields", type=int, default=39)
    ap.add_argument("--embed", type=int, default=64)
    ap.add_argument("--rows-per-gpu", type=int, default=0, help="0 = 1M (N=1) / 125M (N>1)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--zipf", action="store_true", help="Zipf(1.05)-like skewed indices instead of uniform")
    ap.add_argument("--field-layout", default="uniform", choices=["uniform", "skewed"],
                    help="SURVEY 8d's two field-size layouts: uniform (38 x 25 641 + 25 642 at 1 M rows) or criteo-skewed "
                         "(log-spaced sizes from 4 to ~300 k rows summing to the same total: tiny fields = very hot rows)")
    ap.add_argument("--no-other-models", action="store_true",
                    help="skip the short DCN / xDeepFM legs (BASELINE configs[2], [3]) the default one-GPU run appends")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the zipf / skewed / skewed_zipf legs of the headline step the default one-GPU run appends")
    ap.add_argument("--variant-steps", type=int, default=12, help="timed steps of each of those legs")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps of each of those legs")
    ap.add_argument("--other-warmup", type=int, default=2, help="untimed steps in front of them")
    ap.add_argument("--no-fuse", action="store_true", help="separate lookup and FM kernels (drop-in unfused path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-table", action="store_true",
                    help="skip the second roofline entry (the fused lookup+FM launch on a 32 M-row / 4 GiB table)")
    ap.add_argument("--large-table-rows", type=int, default=32_000_000)
    ap.add_argument("--time-every", type=int, default=4,
                    help="bracket every n-th launch of the roofline kernel with HIP events (1 = all launches)")
    ap.add_argument("--force-sharded", action="store_true", help="use the row-sharded path even with one rank (test)")
    ap.add_argument("--shard-graph", default="auto", choices=["auto", "whole", "region", "off"],
                    help="row-sharded runs: what is replayed from a hipGraph.  region: the dense part of the step behind eager "
                         "lookups / exchanges (cross-step pipeline on the communication stream); whole: the entire step, "
                         "exchanges included, as ONE graph (no cross-step overlap: the exchanges run where the step needs "
                         "them); auto = whole on one rank (no wire time to hide; the host leaves the critical path), region "
                         "on more than one")
    ap.add_argument("--model", default="deepfm", choices=["deepfm", "fm", "dcn", "xdeepfm"],
                    help="deepfm = the headline metric (BASELINE configs[1]); dcn / xdeepfm = configs[2] / [3]")
    ap.add_argument("--no-tunableop", action="store_true", help="do not use PyTorch TunableOp for the nn.Linear GEMMs")
    ap.add_argument("--optimizer", default="none", choices=["none", "sgd", "adagrad", "adam"],
                    help="none (default): the metric is fwd+bwd.  sgd/adagrad: also take an optimizer step -- fused sparse "
                         "update inside the embedding backward (no dense table gradient), torch.optim for the MLP")
    ap.add_argument("--microbatches", type=int, default=0,
                    help="sharded path: split each rank's batch into M micro-batches on 2 alternating streams so the "
                         "all-to-all of one overlaps the dense compute of the other (default 1: measured slower on "
                         "one GPU -- per-micro-batch host syncs and sparse-gradient accumulation outweigh the overlap)")
    ap.add_argument("--dedup", action="store_true",
                    help="sharded path: send every distinct row id of the local batch once (pays on skewed indices)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sharded path: issue every exchange where the step needs it (default: the lookup exchange of batch "
                         "k+1 and the gradient exchange of batch k run on a communication stream under the dense compute, "
                         "dist.prefetch_lookup / overlap_grad_exchange)")
    ap.add_argument("--capacity", type=float, default=0.0,
                    help="sharded path: fixed-capacity all-to-all slots (factor on the even share B*N/world, e.g. 1.1): equal "
                         "splits, no split size read on the host (dist.RowShardedMultiIndicesEmbedding(capacity=...))")
    ap.add_argument("--cpu-batch", type=int, default=65536)
    ap.add_argument("--host-indices", action="store_true",
                    help="index batches start in host memory: packed into pinned int32 buffers and copied over PCIe "
                         "inside the timed region (IndexStager), overlapped with the previous step; single-GPU path")
    ap.add_argument("--graph-steps-per-replay", type=int, default=4,
                    help="consecutive steps (each on its own resident batch of the ring) captured into ONE hipGraph; the "
                         "device idles ~25 us between two replays, a K-step graph pays that once per K steps; 1 = one "
                         "step per replay")
    ap.add_argument("--graph-input-sets", type=int, default=1,
                    help="static input sets of the replayed step (torecsys_amd.graph.GraphedStep(input_sets=)): one per "
                         "resident batch of the ring, so a replay copies nothing; 1 = one set, every batch copied into it")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step (forward+backward[+optimizer]) in a hipGraph and replay it; single-GPU path "
                         "(the default for the deepfm / fm workloads on one GPU: the eager step spends 1.3-1.45 ms of "
                         "host time per 1.56 ms step, so its rate depends on the host the driver happens to get)")
    ap.add_argument("--eager", action="store_true", help="never capture: launch every step from Python")
    ap.add_argument("--aten-head", action="store_true",
                    help="A/B: the models' scalar head and BCEWithLogitsLoss as ATen ops (~35 launches) instead of "
                         "functional.ctr_logit / fused.BCEWithLogitsLoss (csrc/head.hip)")
    return ap.parse_args()


def field_sizes(total_rows, n_fields, layout="uniform"):
    if layout == "skewed" and n_fields > 1:
        # log-spaced from 4 rows to 0.3 x total (300 k at 1 M rows), rescaled to the requested total
        import math
        lo, hi = math.log(4.0), math.log(0.3 * total_rows)
        raw = [math.exp(lo + (hi - lo) * i / (n_fields - 1)) for i in range(n_fields)]
        scale = total_rows / sum(raw)
        sizes = [max(4, int(round(x * scale))) for x in raw]
        sizes[-1] += total_rows - sum(sizes)
        assert sizes[-1] >= 4 and sum(sizes) == total_rows
        return sizes
    per = total_rows // n_fields
    return [per] * (n_fields - 1) + [total_rows - per * (n_fields - 1)]


def synth_indices(B, sizes, gen, zipf):
    cols = []
    for f in sizes:
        if zipf:
            r = torch.rand(B, 1, generator=gen, dtype=torch.float64)
            cols.append((torch.pow(float(f), r) - 1.0).clamp_(0, f - 1).long())
        else:
            cols.append(torch.randint(0, f, (B, 1), generator=gen))
    return torch.cat(cols, dim=1)


def cpu_baseline(a, sizes):
    """Oracle DeepFM fwd+bwd (fp32) on a bounded sample of the same workload: one full batch, timed with all host cores
    and with 16 threads (MKL / oneDNN on a many-core box often peak well below the full core count); ``value`` is the
    faster of the two, both are listed.  The oracle is the travelling restatement of the reference's CPU path
    (kind "port"); tools/ref_vs_oracle.py measured reference time / oracle time = 1.04-1.11 in the build container
    (BASELINE.md section 3)."""
    from oracle import cpu_ref as O       # checker / baseline leg only
    B, N, E = a.cpu_batch, a.fields, a.embed
    g = torch.Generator().manual_seed(4321)
    V = sum(sizes)
    off = O.field_offsets(sizes)
    idx = synth_indices(B, sizes, g, a.zipf)
    w = torch.randn(V, E, generator=g).requires_grad_()
    w1 = torch.randn(V, 1, generator=g).requires_grad_()
    dims = [N * E, 400, 400, 400, 1]
    ws = [(torch.randn(o, i, generator=g) / i ** 0.5).requires_grad_() for i, o in zip(dims[:-1], dims[1:])]
    bs = [torch.zeros(o, requires_grad=True) for o in dims[1:]]
    y = (torch.rand(B, 1, generator=g) < 0.25).float()

    def step():
        for t in [w, w1, *ws, *bs]:
            t.grad = None
        emb = O.multi_indices_embedding(w, idx, off)
        feat = O.multi_indices_embedding(w1, idx, off)
        logit = O.deepfm_model(feat, emb, ws, bs)
        nn.functional.binary_cross_entropy_with_logits(logit, y).backward()

    def rate(threads, budget_s):
        torch.set_num_threads(threads)
        step()
        t0 = time.perf_counter()
        n = 0
        while True:
            step()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 8:
                break
        return B * n / el, n

    ncpu = os.cpu_count() or 1
    forced = int(os.environ.get("TRS_CPU_THREADS", "0"))
    runs = {}
    threads_before = torch.get_num_threads()
    for th in ([forced] if forced else sorted({min(16, ncpu), ncpu})):
        runs[th] = rate(th, 6.0)
    # back to what the process ran on: left at all 256 host threads, the small CPU ops of everything behind this leg (the
    # index generation of the variants: 39 x 4 tensors of 65 536 values) crawl -- the default run took 4.5 minutes, 3.5 of
    # them there
    torch.set_num_threads(threads_before)
    best = max(runs, key=lambda t: runs[t][0])
    listing = ", ".join(f"{t} threads: {runs[t][0]:.0f} samples/s ({runs[t][1]} steps)" for t in sorted(runs))
    return {"value": round(runs[best][0], 1), "unit": "samples/s", "cores": best, "kind": "port", "host_cores": ncpu,
            "sample": f"oracle DeepFM fwd+bwd fp32, one batch of {B} of the same synthetic workload (V={V}, {N} fields x "
                      f"dim {E}, MLP [400,400,400]); {listing}"}


def self_check(a, inputs, model, idx, sizes, rows=4096):
    """End-to-end parity AT the benchmark's size and parameters: the logits of the first ``rows`` samples of a timed batch
    from the product path (bf16 kernels) against the fp32 oracle evaluated on the same (bf16-rounded) parameters.
    Checker use of the oracle, part of the cpu_baseline leg."""
    from oracle import cpu_ref as O
    sl = idx[:rows]
    with torch.no_grad():
        d = inputs({"c0": sl})
        got = model(**d).float().cpu()
    sd = {k: v.detach().float().cpu() for k, v in list(inputs.state_dict().items()) + list(model.state_dict().items())}
    off = O.field_offsets(sizes)
    ic = sl.cpu()
    emb = O.multi_indices_embedding(sd["emb_inputs.embedding.weight"], ic, off)
    feat = O.multi_indices_embedding(sd["feat_inputs.embedding.weight"], ic, off)
    if a.model == "deepfm":
        names = [k[:-len(".weight")] for k in sd if k.startswith("deep.model.") and k.endswith(".weight")]
        ref = O.deepfm_model(feat, emb, [sd[n + ".weight"] for n in names], [sd[n + ".bias"] for n in names])
    else:
        ref = O.fm_model(feat, emb, sd.get("bias"))
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    return {"rows": rows, "max_rel_err_logits": round(err, 6), "tolerance": 1e-2 if a.dtype == "bf16" else 1e-5,
            "ok": bool(err <= (1e-2 if a.dtype == "bf16" else 1e-5)),
            "what": "product-path logits of the first rows of timed batch 0 vs the fp32 oracle on the same parameters"}


def _mlp_params(sd, prefix):
    names = [k[:-len(".weight")] for k in sd if k.startswith(prefix) and k.endswith(".weight")]
    return [sd[n + ".weight"] for n in names], [sd[n + ".bias"] for n in names]


def other_model_self_check(name, a, inputs, model, idx, sizes):
    """configs[2] / [3] at the benchmark's size and parameters: product-path logits of the first rows of timed batch 0
    against the fp32 oracle on the same (bf16-rounded) parameters.  DCN: 2048 rows.  xDeepFM: 512 rows (the oracle forms
    the (B, N*H, E) outer product like the reference: 0.65 GB at 512 rows) with BatchNorm in EVAL mode on the running
    statistics the timed training steps left behind -- rows are independent there; the train-mode statistics are pinned
    at this batch size in tests/test_gpu_fullsize.py.  Checker use of the oracle."""
    from oracle import cpu_ref as O
    rows = 2048 if name == "dcn" else 512
    sl = idx[:rows]
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            d = inputs({"c0": sl})
            got = (model(**d) if name != "dcn" else model(emb_inputs=d["emb_inputs"])).float().cpu()
    finally:
        model.train(was_training)
    sd = {k: v.detach().float().cpu() for k, v in list(inputs.state_dict().items()) + list(model.state_dict().items())}
    off = O.field_offsets(sizes)
    ic = sl.cpu()
    emb = O.multi_indices_embedding(sd["emb_inputs.embedding.weight"], ic, off)
    dw, db = _mlp_params(sd, "deep.model.")
    if name == "dcn":
        L = len([k for k in sd if k.startswith("cross.model.") and k.endswith(".weight")])
        ref = O.dcn_model(emb, [sd[f"cross.model.{l}.weight"] for l in range(L)], [sd[f"cross.model.{l}.bias"] for l in range(L)],
                          dw, db, sd["fc.weight"], sd["fc.bias"])
        terms = None
    else:
        feat = O.multi_indices_embedding(sd["feat_inputs.embedding.weight"], ic, off)
        K = len([k for k in sd if k.startswith("cin.model.") and k.endswith(".Conv1d.weight")])
        kw = dict(conv_weights=[sd[f"cin.model.{k}.Conv1d.weight"] for k in range(K)],
                  conv_biases=[sd[f"cin.model.{k}.Conv1d.bias"] for k in range(K)],
                  fc_weight=sd["cin.fc.weight"], fc_bias=sd["cin.fc.bias"],
                  bn_weights=[sd[f"cin.model.{k}.Batchnorm.weight"] for k in range(K)],
                  bn_biases=[sd[f"cin.model.{k}.Batchnorm.bias"] for k in range(K)],
                  bn_running_means=[sd[f"cin.model.{k}.Batchnorm.running_mean"].clone() for k in range(K)],
                  bn_running_vars=[sd[f"cin.model.{k}.Batchnorm.running_var"].clone() for k in range(K)],
                  training=False)
        ref = O.xdeepfm_model(feat, emb, kw, dw, db, sd["bias"])
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    tol = 1e-2 if a.dtype == "bf16" else 1e-5
    return {"rows": rows, "max_rel_err_logits": round(err, 6), "tolerance": tol, "ok": bool(err <= tol),
            "what": "product-path logits of the first rows of timed batch 0 vs the fp32 oracle on the same parameters"
                    + ("" if name == "dcn" else " (BatchNorm in eval mode on the running statistics of the timed steps)")}


WORKLOADS = {
    "deepfm": "BASELINE.json configs[1]: DeepFM 39 Criteo-shaped fields, MLP [400,400,400]",
    "fm": "FactorizationMachine (BASELINE.json configs[0] shape class) on the configs[1] inputs",
    "dcn": "BASELINE.json configs[2]: DeepAndCrossNetwork, 6 cross layers, per-field MLP [400,400,400] -> 64",
    "xdeepfm": "BASELINE.json configs[3]: xDeepFM, CIN [128,128,128], MLP [400,400,400]",
}


MODEL_KERNELS = {"dcn": ["trs_cross_bwd"], "xdeepfm": ["trs_cin16_bwd_data", "trs_cin_cl_bwd_data"]}


def model_kernel_roofline(model, kernel, mtimes, B, N, E, esz):
    """dcn / xdeepfm: the dominant kernel of THEIR step is a matrix-core kernel (cross = MFMA-bound, CIN =
    MFMA-bound): its FLOPs / HIP-event time against the dense bf16 MFMA peak."""
    rows_ = B * N
    if model == "dcn":
        Lc = 6
        flops = 2.0 * rows_ * E * E * (3 * Lc - 1)          # recompute L + gradient chain L-1 + weight gradient L
        per_step = 1
        what = ("cross_mfma_bwd3 (+ prepack and the two partial-sum reductions of the same entry point): "
                "2*rows*E^2*(3L-1) FLOP, detached first layer")
        hbm_alg = 3 * rows_ * E * esz
    else:
        from torecsys_amd import functional as _F
        Hs = [N, 128, 128]
        # channels of gy the contraction runs over: the last layer's "hidden" half carries no gradient and is left out of
        # its backward (functional.CIN_SKIP_DEAD): those FLOPs are NOT executed and are NOT counted
        Cs = [256, 256, 128 if _F.CIN_SKIP_DEAD else 256]
        flops = sum(2.0 * B * E * c * N * h for h, c in zip(Hs, Cs))   # S_n = W_n^T gy over the three layers of a step
        per_step = 3
        what = (kernel.replace("trs_", "") + ", the three layers of a step together: sum_k 2*B*E*C_k*N*H_k FLOP (C_k = "
                + "/".join(str(c) for c in Cs) + ": the last layer's unused hidden half is skipped, not counted; the first "
                "layer runs the symmetric fold and does about half of its share)")
        hbm_alg = None
    tsum = sum(mtimes) / len(mtimes) * per_step * 1e-3       # seconds per step in this kernel
    out = {"bound": "mfma", "kernel": kernel.replace("trs_", ""), "what": what,
           "achieved": round(flops / tsum / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
           "frac": round(flops / tsum / 1e12 / 2500.0, 4), "flops_per_step": flops,
           "us_per_step": round(tsum * 1e6, 1), "launches_timed": len(mtimes)}
    if hbm_alg:
        out["hbm_frac"] = round(hbm_alg / tsum / 1e9 / HBM_PEAK_GBS, 4)
    return out


def _criterion(a):
    """BCEWithLogitsLoss (SURVEY 8d's loss): the HIP one by default (bf16 logits read as they are, 3 launches fwd+bwd),
    ATen's with --aten-head"""
    if a.aten_head:
        return nn.BCEWithLogitsLoss()
    from torecsys_amd.fused import BCEWithLogitsLoss
    return BCEWithLogitsLoss()


def _loss(crit, out, lab):
    return crit(out.float(), lab) if isinstance(crit, nn.BCEWithLogitsLoss) else crit(out, lab)


def other_model_leg(name, a, dev, dt, inputs, idx_ring, label_ring, esz, sizes=None):
    """BASELINE configs[2] / [3] in the default run: a few eager fwd+bwd steps of DCN / xDeepFM on the same inputs,
    same batch ring and same definition of a step as the headline leg (dense table gradients, no optimizer), timed by a
    HIP event pair around the steps; the dominant matrix-core kernel by events around each of its launches."""
    from harness import ctr_models as M
    from torecsys_amd import _abi
    B, N, E = a.batch, a.fields, a.embed
    torch.manual_seed(11)
    if name == "dcn":
        model = M.DeepAndCrossNetworkModel(inputs_size=E, num_fields=N, deep_output_size=64,
                                           deep_layer_sizes=[400, 400, 400], cross_num_layers=6)
    else:
        model = M.XDeepFactorizationMachineModel(embed_size=E, num_fields=N, cin_layer_sizes=[128, 128, 128],
                                                 deep_layer_sizes=[400, 400, 400])
    model = model.to(dev).to(dt)
    crit = _criterion(a)
    params = [p for p in list(inputs.parameters()) + list(model.parameters()) if p.requires_grad]
    ring = len(idx_ring)

    def one(k):
        for p in params:
            p.grad = None
        d = inputs({"c0": idx_ring[k % ring]})
        out = model(**d) if name != "dcn" else model(emb_inputs=d["emb_inputs"])
        loss = _loss(crit, out, label_ring[k % ring])
        loss.backward()
        return loss

    for k in range(max(1, a.other_warmup)):
        one(k)
    torch.cuda.synchronize()
    # The timed steps replay from a hipGraph like the headline leg (one step per replay, the batch copied into the static
    # inputs): launched eagerly a DCN step is ~140 launches and its rate follows the HOST -- 9.2 ms on a warm box, 15-16 ms
    # in the first run on a fresh one (round 6).  Capture refused: the eager loop, as before.
    gstep = None
    if not a.eager and os.environ.get("TRS_OTHER_GRAPH", "1") != "0":
        try:
            from torecsys_amd.graph import GraphedStep

            def gfn(ix, lab):
                d = inputs({"c0": ix})
                out = model(**d) if name != "dcn" else model(emb_inputs=d["emb_inputs"])
                l_ = _loss(crit, out, lab)
                l_.backward()
                return l_

            gstep = GraphedStep(gfn, (idx_ring[0], label_ring[0]), params=params, warmup=1)
        except Exception as exc:      # noqa: BLE001
            print(f"bench.py: hipGraph capture of the {name} step failed ({type(exc).__name__}: {exc}); running eager",
                  file=sys.stderr)
            gstep = None
            torch.cuda.synchronize()
    timed = (lambda k: gstep(idx_ring[k % ring], label_ring[k % ring])) if gstep is not None else one
    kernels = MODEL_KERNELS[name] if dt == torch.bfloat16 else []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import gc
    gc.collect()          # as in the headline leg: a generation-2 pass (~65 ms on the host) inside five 10 ms steps lets the
    gc.disable()          # device run dry -- seen as 19.2 instead of 10.3 ms per DCN step
    timed(0)              # one more untimed step behind that pause: the device idled through it and comes back at a lower
    torch.cuda.synchronize()      # clock (five DCN steps: 9.6 ms each against 9.2 in a 20-step run on the same box)
    e0.record()
    for k in range(a.other_steps):
        loss = timed(k)
    e1.record()
    gc.enable()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.other_steps
    leg = {"workload": WORKLOADS[name], "steps": a.other_steps, "ms_per_step": round(ms, 4),
           "value": round(B / ms * 1e3, 1), "unit": "samples/s", "loss": float(loss.detach()), "hipgraph": gstep is not None}
    if gstep is not None:
        loss = None
        gstep.release_outputs()
    # the dominant matrix-core kernel: sampled in steps of their own, AFTER the timed ones (events around every launch make
    # the host wait on the runtime's profiling signals: they do not belong inside a timed step)
    for kn in kernels:
        _abi.time_kernel(kn, True, expect=3 * a.other_steps + 4, every=1)
    for k in range(min(3, a.other_steps)):
        one(k)
    torch.cuda.synchronize()
    for kn in kernels:
        ts = _abi.kernel_times_ms(kn)
        _abi.time_kernel(kn, False)
        if ts and "roofline_model_kernel" not in leg:
            leg["roofline_model_kernel"] = model_kernel_roofline(name, kn, ts, B, N, E, esz)
    for p in params:
        p.grad = None
    if sizes is not None and not a.no_cpu_baseline:
        try:
            leg["self_check"] = other_model_self_check(name, a, inputs, model, idx_ring[0], sizes)
        except Exception as exc:          # noqa: BLE001 -- the checker must not cost the line
            leg["self_check"] = {"ok": False, "error": f"{type(exc).__name__}: {exc}"}
    del model, gstep
    return leg


def large_table_roofline(a, dev, dt, esz, fm_only=False):
    """The roofline kernel again, stand-alone, on a table that cannot sit in the 256 MiB Infinity Cache (default 32 M
    rows x 64 x bf16 = 4 GiB): same batch shape, HIP events on the launch stream, median of the timed launches.
    ``fm_only``: the (B,N,E) block is not written (want_emb=False) -- north_star's literal "fused embedding + FM forward"
    READ roofline (idx + rows read, FM out written; target <= 73.5 us at the BASELINE shape)."""
    from torecsys_amd import _abi
    from torecsys_amd import functional as F_
    B, N, E = a.batch, a.fields, a.embed
    Vb = a.large_table_rows
    sizes = field_sizes(Vb, N, a.field_layout)
    gen = torch.Generator().manual_seed(99)
    idx = synth_indices(B, sizes, gen, a.zipf).to(dev)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.tensor(sizes), 0)[:-1]]).to(dev)
    w = torch.empty(Vb, E, dtype=dt, device=dev).normal_()
    name = "trs_embed_fm" if not a.no_fuse else "trs_gather_rows"
    fn = ((lambda: F_._EmbedFM.apply(w, idx, off, None, not fm_only)) if not a.no_fuse
          else (lambda: F_._GatherRows.apply(w, idx, off, None)))
    for _ in range(3):
        fn()
    _abi.time_kernel(name, True, expect=12, every=1)
    for _ in range(10):
        fn()
    ts = sorted(_abi.kernel_times_ms(name))
    _abi.time_kernel(name, False)
    del w
    if not ts:
        return None
    med = ts[len(ts) // 2] * 1e-3
    alg = B * N * (8 + E * esz) + (0 if fm_only else B * N * E * esz) + (B * E * esz if not a.no_fuse else 0)
    ach = alg / med / 1e9
    return {"bound": "hbm", "kernel": name.replace("trs_", "") + (" (no block written)" if fm_only else ""),
            "table_rows": Vb, "table_bytes": Vb * E * esz,
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "alg_bytes_per_launch": alg, "median_launch_us": round(med * 1e6, 2), "launches_timed": len(ts)}


def self_launch_argv(gpus, argv, env):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: the command this process replaces itself with --
    one rank per GPU under torch.distributed.run on 127.0.0.1 (the container hostname may not resolve), a free port
    unless MASTER_PORT names one.  None when a launcher already set WORLD_SIZE or N = 1."""
    if gpus <= 1 or "WORLD_SIZE" in env or "RANK" in env:
        return None
    port = env.get("MASTER_PORT")
    if not port:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(argv[0]), *argv[1:]]


_teardown = []       # what main() runs after printing the result line (process-group teardown)

VARIANTS = {"zipf": dict(zipf=True, field_layout="uniform"), "skewed": dict(zipf=False, field_layout="skewed"),
            "skewed_zipf": dict(zipf=True, field_layout="skewed")}


def variant_legs(a):
    """"report both" -- the headline line is quoted on uniform field sizes and uniform indices (the friendliest
    case for the bucket walk's atomics-free reduction and for the caches); these legs time the SAME step (same code path,
    hipGraph replay, resident batches) on Zipf(1.05) indices, on criteo-skewed field sizes (log-spaced, 4 .. ~300 k rows)
    and on both, a few steps each, and report ms per step and the roofline kernel's fraction inside that step."""
    import copy
    out = {}
    for name, kw in VARIANTS.items():
        b = copy.copy(a)
        for k, v in kw.items():
            setattr(b, k, v)
        b.steps, b.warmup = a.variant_steps, max(2, a.variant_steps // 4)
        b.no_cpu_baseline = b.no_large_table = b.no_other_models = b.no_variants = True
        try:
            r = run(b)
            rf = r.get("roofline") or {}
            out[name] = {"workload": r["config"]["workload"], "steps": r["steps"], "ms_per_step": r["ms_per_step"],
                         "value": r["value"], "unit": r["unit"], "loss": r["config"]["loss"],
                         "hipgraph": r["config"]["hipgraph"],
                         "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                             "launches_timed")} if rf else None,
                         "note": "achieved = ALGORITHMIC bytes (SURVEY 8d) / launch time: on skewed ids most rows are served "
                                 "by L2 / the Infinity Cache, not HBM, so the ratio to the HBM peak can pass 1 -- it is a rate "
                                 "of useful bytes, not of HBM traffic"}
        except Exception as exc:      # noqa: BLE001 -- a failed variant must not cost the headline line
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.synchronize()
        from torecsys_amd import functional as _F
        _F.clear_caches()
    return out


def main():
    a = parse()
    relaunch = self_launch_argv(a.gpus, sys.argv, os.environ)
    if relaunch is not None:
        sys.stdout.flush()
        os.execv(relaunch[0], relaunch)
    res = run(a)
    if res is not None:
        # RCCL writes a version banner through C stdio; flush it first so the JSON line is the LAST line
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if _teardown:
        # The process group is torn down AFTER the result is printed, under a deadline: with RCCL all-to-alls captured
        # inside a hipGraph (--shard-graph whole with forced / real collectives) destroy_process_group() was seen to wait
        # for ever on the one-rank communicator (round 6: the run itself had finished -- the stack dump showed the main
        # thread inside destroy_process_group); a rank that hangs there would keep the launcher from returning
        import threading
        killer = threading.Timer(30.0, lambda: os._exit(0))
        killer.daemon = True
        killer.start()
        for fn in _teardown:
            fn()
        killer.cancel()


_T0 = time.perf_counter()


def _stage(name):
    """TRS_BENCH_STAGES=1: wall-clock seconds since start at every stage of the run, on stderr (where the minutes go)"""
    if os.environ.get("TRS_BENCH_STAGES"):
        print(f"[bench stage] {time.perf_counter() - _T0:7.1f} s  {name}", file=sys.stderr, flush=True)


def run(a):
    if os.environ.get("TRS_BENCH_WATCHDOG"):      # developer aid: dump every thread's Python stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["TRS_BENCH_WATCHDOG"]), exit=True, file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    sharded = world > 1 or a.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from torecsys_amd import _abi
    from harness import ctr_models as M
    from torecsys_amd.inputs import Inputs, MultiIndicesEmbedding
    if a.aten_head:
        M.FUSED_HEAD = False

    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    esz = 2 if dt == torch.bfloat16 else 4
    B, N, E = a.batch, a.fields, a.embed
    rows_local = a.rows_per_gpu or (1_000_000 if world == 1 else 125_000_000)
    V = rows_local * world
    sizes = field_sizes(V, N, a.field_layout)
    gen = torch.Generator().manual_seed(1234 + rank)
    # a ring of distinct batches resident in HBM: every step sees new indices, so nothing derived from
    # them (row buckets for the backward) can be reused across steps
    RING = 4
    idx_ring = [synth_indices(B, sizes, gen, a.zipf).to(dev) for _ in range(RING)]
    label_ring = [(torch.rand(B, 1, generator=gen) < 0.25).float().to(dev) for _ in range(RING)]

    _stage("index ring built")
    torch.manual_seed(7)
    pipelined = False
    dense_graph = False
    shard_whole = False
    if not sharded:
        emb = MultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=not a.no_fuse)
        feat = MultiIndicesEmbedding(embed_size=1, field_sizes=sizes)
        parallelism = "single"
    else:
        from torecsys_amd.dist import RowShardedMultiIndicesEmbedding
        if a.capacity == 0.0 and world > 1 and not a.dedup:
            a.capacity = 1.1       # default at N > 1: fixed-capacity slots -- equal all-to-all splits, no split size read on
        cap = a.capacity if a.capacity >= 1.0 else None        # the host (--capacity -1: exact, data-dependent splits)
        mode = a.shard_graph if a.shard_graph != "auto" else ("whole" if world == 1 else "region")
        if a.eager:
            mode = "off"
        # whole: ONE graph per step, lookups and exchanges inside (a captured step must end with every stream joined, so
        # nothing is left running across the step boundary: no cross-step pipeline)
        # (a fused optimizer on a shard beyond the dense-index size compacts the touched rows with torch.unique: a
        # data-dependent shape, not capturable)
        shard_whole = (mode == "whole" and (a.optimizer == "none" or (a.optimizer == "sgd" and rows_local <= 8_000_000))
                       and (a.microbatches or 1) == 1 and not a.dedup
                       and (world == 1 or a.capacity >= 1.0) and a.model in ("deepfm", "fm"))
        pipelined = (not a.no_pipeline and a.optimizer == "none" and (a.microbatches or 1) == 1 and not shard_whole)
        # the dense part of the step (deep branch, head, loss and their backward) replayed from a hipGraph while the
        # lookups and their exchanges stay eager on the compute / communication streams (graph.GraphedRegion)
        dense_graph = (mode == "region" and a.optimizer == "none" and (a.microbatches or 1) == 1 and not a.no_fuse
                       and a.model in ("deepfm", "fm") and not a.dedup)
        emb = RowShardedMultiIndicesEmbedding(embed_size=E, field_sizes=sizes, fuse_fm=not a.no_fuse,
                                              dtype=dt, device=dev, dedup=a.dedup, capacity=cap,
                                              overlap_grad_exchange=pipelined, persistent_outputs=dense_graph)
        feat = RowShardedMultiIndicesEmbedding(embed_size=1, field_sizes=sizes, dtype=dt, device=dev, dedup=a.dedup,
                                               capacity=cap, overlap_grad_exchange=pipelined,
                                               persistent_outputs=dense_graph)
        parallelism = f"row-sharded table x{world} (all-to-all lookup), data-parallel MLP"
    emb.set_schema(["c0"])       # the whole (B,N) index block travels as one named column
    feat.set_schema(["c0"])
    inputs = Inputs(schema={"emb_inputs": emb, "feat_inputs": feat}).to(dev).to(dt)
    if a.model == "deepfm":
        model = M.DeepFactorizationMachineModel(embed_size=E, num_fields=N, deep_layer_sizes=[400, 400, 400],
                                                fm_dropout_p=0.0)
    elif a.model == "fm":
        model = M.FactorizationMachineModel(use_bias=True, dropout_p=0.0)
    elif a.model == "dcn":
        model = M.DeepAndCrossNetworkModel(inputs_size=E, num_fields=N, deep_output_size=64,
                                           deep_layer_sizes=[400, 400, 400], cross_num_layers=6)
    else:
        model = M.XDeepFactorizationMachineModel(embed_size=E, num_fields=N, cin_layer_sizes=[128, 128, 128],
                                                 deep_layer_sizes=[400, 400, 400])
    model = model.to(dev).to(dt)
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    counter = [0]
    crit = _criterion(a)
    params = [p for p in list(inputs.parameters()) + list(model.parameters()) if p.requires_grad]
    dense_opt = None
    if a.optimizer != "none":
        from torecsys_amd.optim import FusedSparseAdagrad, FusedSparseAdam, FusedSparseSGD
        fo = {"sgd": FusedSparseSGD(0.01), "adagrad": FusedSparseAdagrad(0.01), "adam": FusedSparseAdam(1e-3)}[a.optimizer]
        emb.set_fused_optimizer(fo)
        feat.set_fused_optimizer(fo)
        dense_opt = {"sgd": lambda: torch.optim.SGD(model.parameters(), lr=0.01),
                     "adagrad": lambda: torch.optim.Adagrad(model.parameters(), lr=0.01),
                     "adam": lambda: torch.optim.Adam(model.parameters(), lr=1e-3)}[a.optimizer]()

    MB = a.microbatches or 1
    if not sharded:
        MB = 1
    assert B % MB == 0
    mbs = B // MB
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)] if MB > 1 else None

    def fwd_loss(ix, lab, scale):
        d = inputs({"c0": ix})
        out = model(**d) if a.model != "dcn" else model(emb_inputs=d["emb_inputs"])
        loss = _loss(crit, out, lab)
        return loss if scale == 1.0 else loss * scale

    host_idx = a.host_indices and not sharded and MB == 1
    if host_idx:
        from torecsys_amd.staging import IndexStager
        host_ring = [t.cpu().numpy() for t in idx_ring]          # (B,N) int64 arrays in pageable host memory
        stager = IndexStager(B, N, dev, depth=3)
        staged = [stager.stage(host_ring[0])]

    def next_indices(k):
        """index batch of step k: resident in HBM (default) or staged from the host, next batch's copy in flight"""
        if not host_idx:
            return idx_ring[k]
        cur = staged.pop(0).wait()
        staged.append(stager.stage(host_ring[(k + 1) % RING]))
        return cur

    phases = [0.0, 0.0, 0.0, 0]          # host seconds in forward / route prefetch / backward, steps (TRS_BENCH_PHASES=1)
    bucket = None
    if world > 1:          # data-parallel dense parameters: one flat bf16 bucket, all-reduced on the communication stream
        from torecsys_amd.dist import DenseGradBucket
        bucket = DenseGradBucket(model.parameters())
    region = [None]        # graph.GraphedRegion of the dense part (sharded runs), built after the eager warm-up
    lab_static = label_ring[0].clone()
    table_params = [p for p in inputs.parameters() if p.requires_grad]

    def hint_next(k):
        if sharded and pipelined:
            # the NEXT batch's lookup exchange (route, id all-to-all, owner gather, row all-to-all) goes onto the
            # communication stream now and runs under this batch's backward; the gradient exchange of this batch
            # runs under the next forward (overlap_grad_exchange).  fwd+bwd metric: the shards do not change, so the
            # early lookup is bit-identical (tests/test_dist_gloo.py::test_row_sharded_pipelined_step_is_bit_equal)
            emb.prefetch_lookup(idx_ring[(k + 1) % RING])
            feat.prefetch_lookup(idx_ring[(k + 1) % RING])
            if not cap:
                emb.prefetch_route(idx_ring[(k + 3) % RING])      # split sizes read on the host: routed further ahead
        elif sharded:    # input-pipeline style hint: start routing the batch after the next one before this backward
            emb.prefetch_route(idx_ring[(k + 2) % RING])

    def dense_fn(xb, fm_t, ft, lab):
        """the dense part on detached leaves aliasing the lookups' persistent output buffers"""
        xb._trs_fused_fm = (fm_t, xb._version)
        return _loss(crit, model(feat_inputs=ft, emb_inputs=xb), lab)

    def graphed_dense_step(k):
        """lookups eager -> one replay (dense forward + backward) -> gradients back into the lookups, eager"""
        for p in table_params:
            p.grad = None
        d = inputs({"c0": next_indices(k)})
        eo, fo = d["emb_inputs"].rename(None), d["feat_inputs"].rename(None)
        fm_o = d["emb_inputs"]._trs_fused_fm[0]
        if region[0] is None:
            from torecsys_amd.graph import GraphedRegion
            lab_static.copy_(label_ring[k])
            region[0] = GraphedRegion(dense_fn, (eo.detach(), fm_o.detach(), fo.detach(), lab_static),
                                      (True, True, True, False), params=list(model.parameters()), warmup=2)
        lab_static.copy_(label_ring[k], non_blocking=True)
        loss, (g_e, g_fm, g_f, _) = region[0]()
        hint_next(k)
        outs, grads = [], []
        for o_, g_ in ((eo, g_e), (fm_o, g_fm), (fo, g_f)):
            if g_ is not None:
                outs.append(o_)
                grads.append(g_)
        torch.autograd.backward(outs, grads)
        return loss

    dense_ready = [False]      # flipped after the eager warm-up steps (lazy initialisation must not happen in a capture)

    def step():
        k = counter[0] % RING
        counter[0] += 1
        if bucket is not None:
            bucket.wait()       # the previous step's all-reduce still reads / writes the gradients this step replaces
        if not (dense_graph and dense_ready[0]):
            for p in params:
                p.grad = None
        if MB == 1 and dense_graph and dense_ready[0]:
            loss = graphed_dense_step(k)
        elif MB == 1:
            t0 = time.perf_counter()
            loss = fwd_loss(next_indices(k), label_ring[k], 1.0)
            t1 = time.perf_counter()
            hint_next(k)
            t2 = time.perf_counter()
            loss.backward()
            phases[0] += t1 - t0; phases[1] += t2 - t1; phases[2] += time.perf_counter() - t2; phases[3] += 1
            if dense_opt is not None:
                dense_opt.step()
        else:
            # software pipeline over micro-batches: forward(m+1) (routing, all-to-all, gather) overlaps
            # backward(m) on the other stream; backward passes are ordered by events so that gradient
            # accumulation into shared parameters never runs on two streams at once
            main = torch.cuda.current_stream()
            for st in streams:
                st.wait_stream(main)
            prev_bwd = None
            total = None
            for m in range(MB):
                st = streams[m & 1]
                with torch.cuda.stream(st):
                    lm = fwd_loss(idx_ring[k][m * mbs:(m + 1) * mbs], label_ring[k][m * mbs:(m + 1) * mbs], 1.0 / MB)
                    if prev_bwd is not None:
                        st.wait_event(prev_bwd)
                    lm.backward()
                    prev_bwd = torch.cuda.Event()
                    prev_bwd.record(st)
                    total = lm.detach() if total is None else total + lm.detach()
            for st in streams:
                main.wait_stream(st)
            loss = total
        if bucket is not None:      # averaged on the communication stream (dist.DenseGradBucket), under the next forward
            bucket.reduce()
        return loss

    _stage("modules built")
    roof_kernel = "trs_embed_fm" if not a.no_fuse else "trs_gather_rows"
    # dcn / xdeepfm: the dominant kernel of THEIR step is a matrix-core kernel (cross = MFMA-bound, CIN =
    # MFMA-bound): its FLOPs / time against the dense bf16 MFMA peak, next to the fused lookup launch
    model_kernel = MODEL_KERNELS.get(a.model, [None])[-1] if dt == torch.bfloat16 else None
    if a.model == "xdeepfm" and dt == torch.bfloat16:
        from torecsys_amd import layers as _layers_mod
        if getattr(_layers_mod, "CIN_F16", False):
            model_kernel = MODEL_KERNELS["xdeepfm"][0]
    first_kernel = "trs_gather_rows" if (not a.no_fuse and a.model in ("deepfm", "fm", "xdeepfm")) else None
    from torecsys_amd import inputs as _inputs_mod
    if (_inputs_mod.PAIR_FIRST_ORDER and not a.no_fuse and not sharded and a.optimizer == "none"
            and a.model in ("deepfm", "fm")):
        roof_kernel = "trs_embed_fm_fields"     # TRS_PAIR_FIRST_ORDER=1: the first-order lookup rides in the same launch
    want_graph = a.graph or (not a.eager and a.model in ("deepfm", "fm") and not a.host_indices
                             and not os.environ.get("TRS_BENCH_PHASES") and not os.environ.get("TRS_BENCH_CPROFILE"))
    use_graph = (want_graph and MB == 1 and a.optimizer in ("none", "sgd")
                 and ((world == 1 and not sharded) or shard_whole))
    eager_step = step
    for _ in range(a.warmup if not use_graph else max(3, a.warmup // 2)):
        eager_step()
    torch.cuda.synchronize()
    _stage("eager warm-up done")
    if dense_graph:
        try:
            dense_ready[0] = True
            for _ in range(3):
                step()              # the first of these captures the region
            torch.cuda.synchronize()
        except Exception as exc:    # noqa: BLE001 -- capture refused: the eager step is the same work
            print(f"bench.py: hipGraph capture of the dense region failed ({type(exc).__name__}: {exc}); running eager",
                  file=sys.stderr)
            dense_ready[0] = False
            dense_graph = False
            region[0] = None
            torch.cuda.synchronize()
    if os.environ.get("TRS_BENCH_TORCHPROF"):      # developer aid: which host op launches what (3 eager steps)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            eager_step()
            torch.cuda.synchronize()
        with open(os.environ["TRS_BENCH_TORCHPROF"], "w") as f:
            f.write(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60))
            f.write("\n\nhost ops that launch device work, in issue order (op, input shapes -> kernels):\n")
            for ev in sorted(prof.events(), key=lambda e: e.time_range.start):
                ks = getattr(ev, "kernels", None)
                if ks and not any(c.kernels for c in ev.cpu_children if getattr(c, "kernels", None)):
                    f.write(f"{ev.name:44s} {str(ev.input_shapes)[:90]:90s} -> "
                            + ", ".join(f"{k.name[:50]} {k.duration:.1f}us" for k in ks) + "\n")
    if world > 1:
        dist.barrier()
    every = max(1, min(a.time_every, a.steps))

    def enable_kernel_timing():
        """eager steps: HIP events around every `--time-every`-th launch inside the timed region (bracketing all of them
        makes the host wait on the runtime's profiling signals and more than doubles the step, see _abi.time_kernel).
        Replayed steps: the timed graph carries NO timing launches at all; the roofline kernel is sampled after the
        timed region from a second capture of the same step whose launches are bracketed by device timestamp marks."""
        _abi.time_kernel(roof_kernel, True, expect=a.steps // every + 2, every=every)
        if model_kernel:
            _abi.time_kernel(model_kernel, True, expect=3 * a.steps + 8, every=1)
        if first_kernel:       # the E = 1 first-order lookup of the same indices: the other half of SURVEY 8d's unit
            _abi.time_kernel(first_kernel, True, expect=a.steps // every + 2, every=every)

    if not sharded and not use_graph:   # the sharded step reports no single-kernel roofline (alg bytes depend on the routing)
        enable_kernel_timing()
    graph_fn = None
    ksteps_used = [1]
    if use_graph:
        from torecsys_amd.graph import GraphedStep

        one = torch.ones((), dtype=torch.float32, device=dev)      # dL/dL, made once: loss.backward() would fill a fresh one
                                                                   # (a 4.5 us launch) in every step of the graph

        def graph_fn(ix, lab):
            l_ = fwd_loss(ix, lab, 1.0)
            l_.backward(one if l_.dtype == torch.float32 and l_.dim() == 0 else None)
            if dense_opt is not None:
                dense_opt.step()
            return l_

        try:
            # The batches of the ring are resident in HBM before the timed region (the contract's "inputs already resident"):
            # one static input set per ring slot, filled once here -- a replay then reads its batch where it lies, as the
            # eager loop does, instead of copying 20 MB of indices into a single static buffer first (16 us + a launch gap
            # per step; --graph-input-sets 1 restores that).  Batches staged from the host keep the single set: their copy
            # into the static buffer IS the transfer being measured.
            nsets = 1 if host_idx else max(1, min(a.graph_input_sets, RING))
            # --graph-steps-per-replay K (default: the ring length): ONE graph holds K consecutive steps, each on its own
            # resident batch of the ring -- K full forward + backward passes per launch.  Between two replays the device
            # idles for ~25 us (graph-to-graph hand-over: `profiles/r05_bench_deepfm_step_timeline.md`, the gap in front of
            # the next lookup); a K-step graph pays it once per K steps.  Same steps, same order, same final loss.
            kreq = max(1, a.graph_steps_per_replay)
            ksteps = kreq if (not host_idx and (RING % kreq == 0 or kreq % RING == 0) and a.steps % kreq == 0) else 1
            ksteps_used[0] = ksteps
            if ksteps > 1:
                def graph_fn_multi(*flat):
                    for j in range(ksteps):
                        if j:
                            for p in params:
                                p.grad = None          # (host-side: every step of the graph writes fresh gradient tensors)
                        if sharded and j + 1 < ksteps:
                            # the NEXT step's owner bucketing (it depends on its indices only) forks onto the "route" side
                            # stream here and runs beside this step; its forward joins it (dist.prefetch_route)
                            emb.prefetch_route(flat[2 * (j + 1)])
                        l_ = graph_fn(flat[2 * j], flat[2 * j + 1])
                    return l_

                flat = []
                for j in range(ksteps):
                    flat += [idx_ring[j % RING], label_ring[j % RING]]
                gstep = GraphedStep(graph_fn_multi, tuple(flat), params=params, warmup=1)      # static copies of the ring

                def step():
                    k = counter[0]
                    counter[0] += 1
                    if k % ksteps == 0:
                        return gstep.replay(0)          # steps k .. k + ksteps - 1 (ring slots (k + j) % RING)
                    return gstep.output
            else:
                gstep = GraphedStep(graph_fn, (idx_ring[0].int() if host_idx else idx_ring[0], label_ring[0]), params=params,
                                    warmup=1, input_sets=nsets)      # staged batches arrive as int32
                if nsets > 1:
                    assert RING % nsets == 0
                    for k in range(nsets):
                        gstep.load(k, idx_ring[k], label_ring[k])
                    torch.cuda.synchronize()

                def step():
                    k = counter[0] % RING
                    counter[0] += 1
                    if nsets > 1 and k < nsets and nsets == RING:
                        return gstep.replay(k)                        # the batch is already where the graph reads it
                    return gstep(next_indices(k), label_ring[k])      # copies the batch into the static buffers, replays

            for _ in range(a.warmup):
                step()
        except Exception as exc:                   # capture refused on this box / runtime: the eager step is the same work
            if a.graph:
                raise
            print(f"bench.py: hipGraph capture failed ({type(exc).__name__}: {exc}); running eager", file=sys.stderr)
            use_graph = False
            step = eager_step
            torch.cuda.synchronize()
            enable_kernel_timing()
            for _ in range(a.warmup):
                step()
            _abi.kernel_times_ms(roof_kernel)
            if first_kernel:
                _abi.kernel_times_ms(first_kernel)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    phases[:] = [0.0, 0.0, 0.0, 0]
    if not host_idx:
        counter[0] = 0      # the timed steps walk the batch ring from its start in every mode (eager / replayed): same final loss
    # no cyclic-GC pass inside the timed region: a generation-2 collection over the imported torch modules takes
    # ~65 ms here, i.e. tens of steps (seen as one 65 ms step in the row-sharded run); objects are freed by refcount
    import gc
    gc.collect()
    gc.disable()
    # ... and a collection pass idles the device for those ~65 ms: it comes back at a lower clock, which a 20-step timed
    # region (24 ms) would carry in full.  A few more untimed steps behind the pause (a whole number of ring rounds and of
    # graph replays, so the timed steps still start at ring slot 0), then the synchronisation the contract asks for.
    rewarm = max(RING, ksteps_used[0])
    if not host_idx and rewarm % RING == 0 and rewarm % ksteps_used[0] == 0:
        for _ in range(rewarm):
            step()
        if not sharded and not use_graph:      # eager steps carry the sampled launches: these samples are not the region's
            for kn in (roof_kernel, first_kernel, model_kernel):
                if kn:
                    _abi.kernel_times_ms(kn)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        counter[0] = 0
        phases[:] = [0.0, 0.0, 0.0, 0]
    _stage("capture and re-warm done")
    dev_allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    prof = None
    if os.environ.get("TRS_BENCH_CPROFILE"):      # developer diagnostic: which host call blocks inside the timed region
        import cProfile
        prof = cProfile.Profile()
    span0, span1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    span0.record()          # device-side cross-check of the wall clock (one event pair around the whole region)
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    stamps = []
    for i in range(a.steps):
        loss = step()
        stamps.append(time.perf_counter())
    if prof:
        prof.disable()
    enqueue_s = time.perf_counter() - t0
    span1.record()
    gc.enable()
    torch.cuda.synchronize()
    span1.synchronize()      # belt and braces: the region's last event has completed
    final_loss = float(loss.detach())
    if bucket is not None:
        bucket.wait()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if prof:
        import pstats
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    if os.environ.get("TRS_BENCH_PHASES") and phases[3]:
        print("host ms/step  forward %.3f  prefetch %.3f  backward %.3f  (over the %d timed eager steps)" %
              tuple([1e3 * v / phases[3] for v in phases[:3]] + [phases[3]]), file=sys.stderr)
        print("host ms per step call:", " ".join("%.2f" % ((b_ - a_) * 1e3) for a_, b_ in zip([t0] + stamps, stamps)),
              file=sys.stderr)
        print("device allocations (hipMalloc) inside the timed region:",
              torch.cuda.memory_stats().get("num_device_alloc", 0) - dev_allocs0,
              " reserved GB: %.2f" % (torch.cuda.memory_reserved() / 2**30), file=sys.stderr)
        if sharded:
            from torecsys_amd import dist as _d
            print("route plans:", _d.route_stats, file=sys.stderr)
        # how fast is THIS box: a dense bf16 GEMM and a device copy (boxes of the pool differ in sustained clocks)
        ga = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        gb_ = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        src_ = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
        dst_ = torch.empty_like(src_)
        for _ in range(3):
            ga @ gb_
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(20):
            ga @ gb_
        e1.record()
        for _ in range(20):
            dst_.copy_(src_)
        e2.record()
        torch.cuda.synchronize()
        print("box probe: bf16 GEMM 8192^3 %.0f TFLOP/s, device copy %.0f GB/s (read+write)" %
              (20 * 2 * 8192 ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12,
               20 * 2 * (1 << 28) / (e1.elapsed_time(e2) * 1e-3) / 1e9), file=sys.stderr)
    _stage("timed region done")
    device_span_ms = span0.elapsed_time(span1)
    shard_diag = None
    if sharded:
        # diagnostic leg, outside the timed region: a few more steps with a HIP event pair around every phase of the
        # sharded lookup (on the stream it runs on) and the bytes each rank puts on the wire
        from torecsys_amd import dist as _d
        _d.PROFILE = True
        _d.phase_events.clear()
        for kk in _d.wire_bytes:
            _d.wire_bytes[kk] = 0
        if use_graph:             # phase events cannot be read out of a replayed graph: the same step, eagerly, for this leg
            loss = None
            gstep.release_outputs()
        for _ in range(6):
            (eager_step if use_graph else step)()
        torch.cuda.synchronize()
        ph = _d.phase_times_ms()
        _d.PROFILE = False
        nst = max(1, _d.wire_bytes["steps"] // 2)          # two sharded tables (E = 64 and E = 1) per step
        shard_diag = {"phase_ms_per_call": {k_: round(v_, 4) for k_, v_ in ph.items()},
                      "wire_bytes_sent_per_step": {k_: int(v_ / nst) for k_, v_ in _d.wire_bytes.items() if k_ != "steps"},
                      "pipelined": bool(pipelined), "lookups": dict(_d.lookup_stats), "routes": dict(_d.route_stats),
                      "note": "device ms per call of each phase (two tables per step share the route); rank 0 only"}
    sampled_replays = 0
    if use_graph and not sharded:
        # the roofline kernel inside the step, sampled AFTER the timed region: a second capture of the same step with its
        # launches of the roofline kernel (and of the first-order lookup) bracketed by two device timestamp marks each;
        # every replay adds one sample per kernel.  The timed replays above carried none of these launches.
        enable_kernel_timing()
        torch.cuda.synchronize()
        loss = None
        gstep.release_outputs()   # the first capture's autograd graphs must be gone before the parameters are captured again
        gstep_t = GraphedStep(graph_fn, (idx_ring[0].int() if host_idx else idx_ring[0], label_ring[0]), params=params,
                              warmup=1)
        sampled_replays = max(8, min(32, a.steps))
        for j in range(sampled_replays):
            gstep_t(idx_ring[j % RING].int() if host_idx else idx_ring[j % RING], label_ring[j % RING])
        torch.cuda.synchronize()
    ktimes = _abi.kernel_times_ms(roof_kernel)
    _abi.time_kernel(roof_kernel, False)
    mtimes = _abi.kernel_times_ms(model_kernel) if (model_kernel and not sharded) else []
    ftimes = _abi.kernel_times_ms(first_kernel) if (first_kernel and not sharded) else []
    for kn in (model_kernel, first_kernel):
        if kn:
            _abi.time_kernel(kn, False)
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        # algorithmic bytes of one fused lookup+FM forward launch over the rows this rank gathers
        # (per sample N*(8 + E*s) read, E*s (+ N*E*s block) + 4*E (fp32 sum) written)
        # idx B*N*8 + rows B*N*E*s read, FM out B*E*s (+ the block B*N*E*s) written.  The fp32
        # field sums the kernel also leaves for its backward (B*E*4) are an implementation side output: listed, not counted
        if not a.no_fuse:
            alg = B * N * (8 + E * esz) + B * N * E * esz + B * E * esz
            side = B * E * 4
        else:
            alg = B * N * (8 + E * esz) + B * N * E * esz
            side = 0
        if sharded:
            alg = None
        kt = sum(ktimes) / max(1, len(ktimes)) * 1e-3 if ktimes else None
        traffic = traffic_round = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                import hashlib
                src = os.path.join(ROOT, "torecsys_amd", "csrc", "fm.hip")
                # a figure taken from another build of the kernel is not this kernel's traffic: refused
                # (and the figure belongs to the configuration it was profiled on: BASELINE configs[1], uniform field sizes)
                if (tj.get("kernel_source_sha16") == hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
                        and a.field_layout == "uniform" and a.model == "deepfm" and B == 65536):
                    traffic = tj.get(roof_kernel + ("_zipf" if a.zipf else ""))
                    traffic_round = tj.get("round")
            except Exception:  # noqa: BLE001
                traffic = None
        roof = None
        if kt and alg:
            ach = alg / kt / 1e9
            roof = {"bound": "hbm", "kernel": roof_kernel.replace("trs_", ""), "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_round": traffic_round,
                    "traffic_source": "profiles/traffic.json: HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / "
                                      "WRITE_SIZE passes of this command (tools/pmc_traffic.sh), corrected per "
                                      "MI355X_MICROARCH.md (1 KiB units, FETCH_SIZE x 2 on gfx950); counters cannot be read "
                                      "inside a timed run, so the figure is the committed one of the same kernel source "
                                      "(hash-checked: null when fm.hip differs from the profiled build)",
                    "alg_bytes_per_launch": alg, "side_output_bytes": side, "avg_launch_us": round(kt * 1e6, 2),
                    "launches_timed": len(ktimes),
                    "timed_how": ("device timestamp marks around the launch in %d replays of a second capture of the step, "
                                  "taken after the timed region (the timed replays carry no timing launches)" % sampled_replays)
                                 if sampled_replays else "HIP events around every %d-th launch inside the timed region" % every,
                    "note": f"the {V * E * esz >> 20} MiB table of this configuration fits the 256 MiB Infinity Cache; "
                            "roofline_large_table repeats the same launch on a table that does not"}
        big = big_fm = None
        if roof is not None and not a.no_large_table and world == 1:
            _stage("timed region and kernel sampling done")
            big = large_table_roofline(a, dev, dt, esz)
            if not a.no_fuse:
                big_fm = large_table_roofline(a, dev, dt, esz, fm_only=True)
        # SURVEY 8d's FULL unit: fused lookup + FM of the E-wide table AND the first-order rows of the same indices
        # (B*N*s more bytes read) -- two launches in this step (folding the E = 1 lookup into the wide kernel was built
        # and measured slower): bytes of both / time of both
        full_unit = None
        if roof is not None and ftimes and kt:
            ft = sum(ftimes) / len(ftimes) * 1e-3
            alg_full = alg + B * N * esz + B * N * esz            # first-order rows read (+ the (B,N,1) values written)
            full_unit = {"bound": "hbm", "kernels": [roof_kernel.replace("trs_", ""), "gather_rows (E=1 first-order table)"],
                         "achieved": round(alg_full / (kt + ft) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_full / (kt + ft) / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes": alg_full,
                         "avg_launch_us": [round(kt * 1e6, 2), round(ft * 1e6, 2)]}
        model_roof = model_kernel_roofline(a.model, model_kernel, mtimes, B, N, E, esz) if mtimes else None
        res = {
            "metric": "CTR samples/sec fwd+bwd (DeepFM, 39 fields x dim 64)" if a.model == "deepfm" else
                      f"CTR samples/sec fwd+bwd ({a.model}, 39 fields x dim 64)",
            "value": round(B * world * a.steps / el, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (WORKLOADS[a.model] + f", {V} total rows, embed_dim {E}, batch {B}, "
                                    + ("zipf" if a.zipf else "uniform") + f" indices, {a.field_layout} field sizes") if world == 1 else
                       ("BASELINE.json configs[4] (weak-scaled): DeepFM, " f"{V} rows row-sharded over {world} GPUs, "
                        f"global batch {B * world}"),
                       "global_batch": B * world, "rows": V, "parallelism": parallelism,
                       "microbatches": MB, "optimizer": a.optimizer, "hipgraph": bool(use_graph or (dense_graph and dense_ready[0])),
                       "hipgraph_input_sets": (nsets if use_graph else None),
                       "hipgraph_steps_per_replay": (ksteps if use_graph else None),
                       **({"hipgraph_scope": "dense region (deep branch, head, loss, their backward) replayed; lookups, "
                                             "exchanges and the dense all-reduce eager on the compute / communication streams"}
                          if (dense_graph and dense_ready[0]) else {}),
                       **({"hipgraph_scope": "whole step (lookups, exchanges, dense part, backward) in one graph; no "
                                             "cross-step overlap of the exchanges"} if (sharded and use_graph) else {}),
                       "indices_from_host": bool(host_idx),
                       "fused_lookup_fm": not a.no_fuse, "loss": final_loss,
                       "host_enqueue_ms_per_step": round(enqueue_s / a.steps * 1e3, 4),
                       "device_span_ms_per_step": round(device_span_ms / a.steps, 4),
                       **({"sharded": shard_diag} if shard_diag is not None else {})},
            "roofline": roof,
        }
        if big is not None:
            res["roofline_large_table"] = big
        if big_fm is not None:
            res["roofline_fm_only_large_table"] = big_fm
        if full_unit is not None:
            res["roofline_full_unit"] = full_unit
        if model_roof is not None:
            res["roofline_model_kernel"] = model_roof
        # GPU legs first, the CPU baseline LAST: its 256-thread run leaves the host's OpenMP pool spinning, which slows the
        # Python threads that enqueue the eager DCN / xDeepFM steps (seen as 15 instead of 9 ms per DCN step) and the small
        # CPU ops of the variants' index generation (the default run took 4.5 minutes with the baseline in front)
        _stage("large-table rooflines done")
        if (world == 1 and not sharded and a.model == "deepfm" and not a.no_other_models and a.optimizer == "none"
                and not a.no_fuse):
            # BASELINE configs[2] and [3] ride along: a few steps each, reported beside the headline (never as `value`)
            res["other_models"] = {m: other_model_leg(m, a, dev, dt, inputs, idx_ring, label_ring, esz, sizes)
                                   for m in ("dcn", "xdeepfm")}
            _stage("dcn / xdeepfm legs done")
        if (world == 1 and not sharded and a.model == "deepfm" and not getattr(a, "no_variants", False)
                and not a.no_other_models and a.optimizer == "none" and not a.no_fuse and not a.zipf and a.field_layout == "uniform"
                and not a.host_indices):
            res["variants"] = variant_legs(a)
            _stage("variants done")
        if world == 1 and not a.no_cpu_baseline:
            if not sharded and a.optimizer == "none" and a.model in ("deepfm", "fm"):
                res["self_check"] = self_check(a, inputs, model, idx_ring[0], sizes)
                _stage("self check done")
            res["cpu_baseline"] = cpu_baseline(a, field_sizes(1_000_000, N, a.field_layout))
            _stage("cpu baseline done")
    else:
        res = None
    if sharded:
        _teardown.append(dist.destroy_process_group)      # after the JSON line is out: see main()
    return res


if __name__ == "__main__":
    main()
