#!/usr/bin/env python3
"""Developer probe: GPU busy time (union of kernel intervals) per step from a rocprofv3 --kernel-trace sqlite result.
usage: tools/gpu_busy.py <results.db> <steps-in-trace> [marker-kernel-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else "embed_fm_kernel"
try:
    rows = c.execute("select start, end, name from kernels order by start").fetchall()
except sqlite3.Error:
    print([r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")])
    raise
marks = [s for s, e, n in rows if marker in n]
print("dispatches", len(rows), "marker launches", len(marks))
if len(marks) >= steps + 1:
    # the last `steps` full steps: from the marker of step -steps-1 ... to the last marker
    lo, hi = marks[-steps - 1], marks[-1]
    sel = [(s, e) for s, e, n in rows if lo <= s < hi]
    busy, cur_s, cur_e = 0, None, None
    for s, e in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e in sel)
    gaps = sorted(sel[i + 1][0] - max(x[1] for x in sel[:i + 1][-4:]) for i in range(len(sel) - 1))
    print("per step over the last %d steps: span %.3f ms  busy(union) %.3f ms  sum of durations %.3f ms  kernels %.1f" %
          (steps, (hi - lo) / steps / 1e6, busy / steps / 1e6, tot / steps / 1e6, len(sel) / steps))
