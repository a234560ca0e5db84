#!/usr/bin/env python3
"""profiles/traffic.json from the two counter passes of tools/pmc_traffic.sh (HBM bytes per launch of the roofline kernel,
corrected as MI355X_MICROARCH.md prescribes: counter units of 1 KiB, FETCH_SIZE x 2 on gfx950), stamped with the round and
the hash of the kernel's source so that bench.py can refuse a figure taken from another build.
usage: tools/make_traffic_json.py <dir with pmc_FETCH_SIZE.txt / pmc_WRITE_SIZE.txt> <round> [--md profiles/rNN_pmc_embed_fm.md]"""
import argparse
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_hash():
    return hashlib.sha256(open(os.path.join(ROOT, "torecsys_amd", "csrc", "fm.hip"), "rb").read()).hexdigest()[:16]


def parse(path):
    """{kernel: (dispatches, value)} of a tools/pmc_summary.py listing"""
    out, cur, n = {}, None, 0
    for line in open(path):
        m = re.match(r"(\S.*?)\s+\(dispatches (\d+)\)", line)
        if m:
            cur, n = m.group(1), int(m.group(2))
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)", line)
        if m and cur:
            out[cur] = (n, float(m.group(2)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("round", type=int)
    ap.add_argument("--md", default=None)
    ap.add_argument("--cmd", default="python bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 10 --warmup 3 --eager")
    a = ap.parse_args()
    f, w = parse(os.path.join(a.dir, "pmc_FETCH_SIZE.txt")), parse(os.path.join(a.dir, "pmc_WRITE_SIZE.txt"))
    res = {"round": a.round, "kernel_source_sha16": kernel_hash(), "command": a.cmd,
           "_note": "bytes per launch at B=65536,N=39,E=64,V=1M,bf16: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM section)"}
    rows = []
    for k in sorted(f):
        if k not in w:
            continue
        rd, wr = 2 * f[k][1] * 1024, w[k][1] * 1024
        rows.append((k, f[k][0], f[k][1], w[k][1], rd, wr))
        if "embed_fm_group_kernel" in k and "true" in k:
            res["trs_embed_fm"] = int(rd + wr)
        if "scatter_rows_fm1_kernel" in k:
            res["trs_scatter_rows_fm1"] = int(rd + wr)
    json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    if a.md:
        lines = [f"# Round {a.round} -- HBM traffic of the lookup + FM kernel and the bucket walk, from the PMC counters", "",
                 "commands (one counter per pass, counters-only -- `tools/pmc_traffic.sh`):", "",
                 f"    rocprofv3 --pmc FETCH_SIZE --kernel-trace -- {a.cmd}", f"    rocprofv3 --pmc WRITE_SIZE --kernel-trace -- {a.cmd}", "",
                 "means per launch; bytes per MI355X_MICROARCH.md (counter units of 1 KiB; FETCH_SIZE x 2 on gfx950); "
                 f"kernel source `fm.hip` sha256[:16] = {res['kernel_source_sha16']}", "",
                 "| kernel | launches | FETCH_SIZE | WRITE_SIZE | bytes read | bytes written | total per launch |", "|---|---:|---:|---:|---:|---:|---:|"]
        for k, n, fv, wv, rd, wr in rows:
            lines.append(f"| `{k[:90]}` | {n} | {fv:.1f} | {wv:.1f} | {rd / 1e6:.1f} MB | {wr / 1e6:.1f} MB | **{(rd + wr) / 1e6:.1f} MB** |")
        lines += ["", "algorithmic bytes of `embed_fm` (SURVEY 8d): 683.1 MB + the 16.8 MB fp32 `fm_sum` side output = 699.9 MB."]
        open(a.md, "w").write("\n".join(lines) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
