"""Assemble profiles/<tag>_kernels.md from one sweep of tools/run_round_measurements.sh (gpurun_out/<tag>/).

usage: python tools/make_kernels_md.py r02
"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)


def line(name):
    with open(os.path.join(src, name)) as f:
        return json.loads(f.read().strip().split("\n")[-1])


rows = [("DeepFM (default bench line: step replayed from a hipGraph)", "bench_deepfm.json"),
        ("DeepFM, launched eagerly (`--eager`)", "bench_deepfm_eager.json"),
        ("DeepFM, Zipf(1.05) indices", "bench_deepfm_zipf.json"),
        ("DeepFM, criteo-skewed field sizes (4 ... 256 k rows), uniform indices", "bench_deepfm_skewed.json"),
        ("DeepFM, criteo-skewed field sizes, Zipf(1.05) indices", "bench_deepfm_skewed_zipf.json"),
        ("DeepFM + fused sparse Adagrad (eager)", "bench_deepfm_adagrad.json"),
        ("DeepFM, row-sharded path on ONE rank, 1 M rows (whole step in one hipGraph; own lookups straight from the shard)", "bench_deepfm_sharded1.json"),
        ("same, the 125 M-row x 64 shard of BASELINE configs[4] (16 GB; sparse COO gradient)", "bench_deepfm_sharded1_125m.json"),
        ("one rank, 1 M rows, the arrangement of a larger world: dense region replayed, lookups / exchanges eager and pipelined (`--shard-graph region`)", "bench_deepfm_sharded1_region.json"),
        ("same, 125 M rows", "bench_deepfm_sharded1_region_125m.json"),
        ("one rank, 1 M rows, EVERY row through the gather / exchange buffers (`TRS_SHARD_LOCAL_DIRECT=0`: what a rank of an 8-GPU world does for the 7/8 of its lookups that other ranks own)", "bench_deepfm_sharded1_buffers.json"),
        ("same, 125 M rows", "bench_deepfm_sharded1_buffers_125m.json"),
        ("one rank, 1 M rows + fused Adagrad on the owner (eager)", "bench_deepfm_sharded1_adagrad.json"),
        ("same, 125 M rows (compact-row optimizer path)", "bench_deepfm_sharded1_adagrad_125m.json"),
        ("FM (replayed)", "bench_fm.json"),
        ("DCN x6", "bench_dcn.json"),
        ("xDeepFM CIN [128,128,128]", "bench_xdeepfm.json")]
out = [f"# Round {int(tag[1:])} -- bench lines and stand-alone kernel timings of one sweep (`tools/run_round_measurements.sh {tag}`)",
       "", "One MI355X box, one sweep (box-to-box and run-to-run spread is 2-5 %).", "",
       "| configuration | ms / step | M samples/s | roofline.frac (lookup+FM kernel, in step) |", "|---|---:|---:|---:|"]
for label, name in rows:
    try:
        d = line(name)
    except (OSError, ValueError):
        continue
    frac = (d.get("roofline") or {}).get("frac")
    out.append(f"| {label} | {d['ms_per_step']:.3f} | {d['value'] / 1e6:.2f} | {frac if frac is not None else ''} |")
d = line("bench_deepfm.json")
lt = d.get("roofline_large_table") or {}
cb = d.get("cpu_baseline") or {}
om = d.get("other_models") or {}
if om:
    out += ["", "`other_models` legs of the default line (5 eager steps each, same inputs): " +
            "; ".join(f"{k}: {v['ms_per_step']:.2f} ms/step, model kernel {v.get('roofline_model_kernel', {}).get('kernel')} "
                      f"{v.get('roofline_model_kernel', {}).get('achieved')} TFLOP/s" for k, v in om.items()) + "."]
for label, name in (("1 M rows, own lookups straight from the shard", "bench_deepfm_sharded1.json"),
                    ("125 M rows, own lookups straight from the shard", "bench_deepfm_sharded1_125m.json"),
                    ("1 M rows, every row through the buffers", "bench_deepfm_sharded1_buffers.json"),
                    ("125 M rows, every row through the buffers", "bench_deepfm_sharded1_buffers_125m.json")):
    if not os.path.exists(os.path.join(src, name)):
        continue
    try:
        c = line(name).get("config") or {}
    except ValueError:
        continue
    sh = c.get("sharded")
    if sh:
        out += ["", f"One-rank sharded step ({label}; host enqueue {c.get('host_enqueue_ms_per_step')} ms/step), device ms per phase "
                "call, per table (eager diagnostic leg outside the timed region): " +
                ", ".join(f"{k} {v}" for k, v in sh["phase_ms_per_call"].items()) + "."]
out += ["", f"Default line extras: `roofline_large_table.frac` = {lt.get('frac')} ({lt.get('rows', '32 M')}-row table), "
        f"`cpu_baseline` = {cb.get('value')} {cb.get('unit', 'samples/s')} on {cb.get('cores')} threads ({cb.get('sample', '')}).",
        "", "The default line as printed:", "", "```", json.dumps(d), "```", "",
        "## stand-alone kernels (`tools/kbench.py`, HIP events on the launch stream; B = 65 536, N = 39, E = 64, bf16 unless noted)",
        "", "```"]
for name in ("kbench.txt", "kbench_pairx_mlpf.txt", "kbench_ffm.txt"):
    with open(os.path.join(src, name)) as f:
        out += [ln.rstrip() for ln in f if ln.strip() and "amdgpu.ids" not in ln]
out += ["```", ""]
dst = os.path.join(root, "profiles", f"{tag}_kernels.md")
with open(dst, "w") as f:
    f.write("\n".join(out))
print("wrote", dst)
