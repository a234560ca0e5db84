"""Print VGPR / spill / occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

usage: python tools/regs.py torecsys_amd/csrc/cin_mfma.hip [name-filter]
"""
import re
import subprocess
import sys
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
sys.path.insert(0, root)
from torecsys_amd.build import FILE_FLAGS  # noqa: E402
cmd = ["hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-ffp-contract=off", *FILE_FLAGS.get(os.path.basename(src), []),
       f"-I{root}/include",
       f"-I{root}/torecsys_amd/csrc", "-c", src, "-o", "/tmp/_regs.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if flt in name:
        print(f"{name:70s} vgpr {r.get('VGPRs'):>4} agpr {r.get('AGPRs'):>4} spill {r.get('VGPRs Spill'):>4} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
