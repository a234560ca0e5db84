"""Build libtrs_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m torecsys_amd.build [--force] [--jobs N]

hipcc cross-compiles without a GPU.  Objects are compiled with hipcc (device code embedded) and
linked with g++ against the libamdhip64 that PyTorch itself loads, so the process holds exactly one
HIP runtime (streams / device pointers are shared with torch).
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtrs_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++20", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-result"]
# per-file additions.  The SLP vectoriser turns fp32 arithmetic that sits between MFMAs into v_pk_* instructions, which
# cost more issue time beside the matrix pipe than the two single ones they replace (CIN forward 1.58 vs 1.18 ms, cross
# backward 1.15 vs 1.04 ms)
FILE_FLAGS = {"cin_mfma.hip": ["-fno-slp-vectorize"], "cross_mfma.hip": ["-fno-slp-vectorize"],
              "mlp_fused.hip": ["-fno-slp-vectorize"], "mlp_ro.hip": ["-fno-slp-vectorize"],
              **{f"mlp_ro_{k}.hip": ["-fno-slp-vectorize"] for k in ("dcn_fwd", "dcn_bwd", "tail_fwd", "tail_bwd")}}


# developer A/B: TRS_BUILD_SLP="cross_mfma.hip,cin_mfma.hip" compiles those files WITH the SLP vectoriser (packed fp32 math)
for _f in filter(None, os.environ.get("TRS_BUILD_SLP", "").split(",")):
    FILE_FLAGS[_f] = [x for x in FILE_FLAGS.get(_f, []) if x != "-fno-slp-vectorize"]


# developer A/B: TRS_BUILD_DEFS="cross_mfma.hip:-DTRS_B3_BARRIER" adds compiler options to single files
for _e in filter(None, os.environ.get("TRS_BUILD_DEFS", "").split(",")):
    _f, _, _d = _e.partition(":")
    FILE_FLAGS[_f] = FILE_FLAGS.get(_f, []) + [_d]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "trs_abi.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(CSRC, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime())):
        return obj, ""
    cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def build(force=False, jobs=None, verbose=False):
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=jobs or min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    if verbose:
        for _, err in results:
            if err.strip():
                print(err, file=sys.stderr)
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        cmd = ["g++", "-shared", "-o", OUT, *objs, "-L" + torch_lib_dir(), "-lamdhip64"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.jobs, a.verbose))
