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
        ("DeepFM, row-sharded path on one rank (pipelined exchanges: nothing to overlap on one rank)", "bench_deepfm_sharded1.json"),
        ("same, exchanges in program order (`--no-pipeline`)", "bench_deepfm_sharded1_nopipe.json"),
        ("same + fused Adagrad on the owner", "bench_deepfm_sharded1_adagrad.json"),
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
sh = (line("bench_deepfm_sharded1.json").get("config") or {}).get("sharded") if os.path.exists(os.path.join(src, "bench_deepfm_sharded1.json")) else None
if sh:
    out += ["", "One-rank sharded step, device ms per phase call (diagnostic leg outside the timed region): " +
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
