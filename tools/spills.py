"""VGPR spills (scratch memory traffic) of every kernel of torecsys_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage,
the build's own flags).  SGPR spills are not counted: they live in lanes of a VGPR, not in memory.  Prints one line per
kernel that spills; exit code 1 if any does.

    python tools/spills.py [file.hip ...]
"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torecsys_amd.build import ARCH, CSRC, FILE_FLAGS, FLAGS, HIPCC, sources  # noqa: E402


def analyse(src):
    path = os.path.join(CSRC, src)
    cmd = [HIPCC, *[f for f in FLAGS if f != "-fPIC"], *FILE_FLAGS.get(src, []), "-c", path, "-o", os.devnull,
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
    rows, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = v
            rows[cur] = {}
        elif cur:
            rows[cur][k] = v
    return src, rows


def demangle(names):
    if not names:
        return {}
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def spilling_kernels(files=None):
    files = files or sources()
    bad = []
    with cf.ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        for src, rows in ex.map(analyse, files):
            names = demangle(list(rows))
            for k, r in rows.items():
                vs, ss = int(r.get("VGPRs Spill", 0)), int(r.get("SGPRs Spill", 0))
                if vs:
                    bad.append((src, re.sub(r"\(.*", "", names.get(k, k)), vs, ss, r.get("VGPRs"), r.get("AGPRs")))
    return bad


if __name__ == "__main__":
    bad = spilling_kernels([os.path.basename(f) for f in sys.argv[1:]] or None)
    for src, name, vs, ss, v, a in sorted(bad):
        print(f"{src:18s} {name:90s} vgpr spill {vs:4d} sgpr spill {ss:3d} (vgpr {v} agpr {a})")
    print(f"{len(bad)} spilling kernels")
    sys.exit(1 if bad else 0)
