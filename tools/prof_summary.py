#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats) sqlite result into a compact per-kernel summary (markdown).
usage: tools/prof_summary.py <results.db> [--out profiles/xxx.md] [--title "..."] [--cmd "..."]"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if name.startswith("Cijk_") or name.startswith("Custom_Cijk"):
        m = re.search(r"(MT\d+x\d+x\d+)", name)
        return "hipBLASLt GEMM " + name[:22] + ".." + (m.group(1) if m else "")
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--out", default=None)
    ap.add_argument("--title", default="rocprofv3 --kernel-trace --stats summary")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--calls", default=None, help="regex: also list the last --ncalls launches of matching kernels in launch order")
    ap.add_argument("--ncalls", type=int, default=40)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = [f"# {a.title}", ""]
    if a.cmd:
        lines += [f"command: `{a.cmd}`", ""]
    lines += ["durations in microseconds (rocprofv3 `top_kernels` view)", "",
              "| kernel | calls | total (us) | avg (us) | % |", "|---|---:|---:|---:|---:|"]
    tot = sum(r[2] for r in rows)
    for name, calls, total, avg, pct in rows:
        lines.append(f"| `{short(name)}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")
    lines.append(f"| **all kernels** | {sum(r[1] for r in rows)} | {tot:.1f} | | 100 |")
    if a.calls:
        seq = c.execute("select name, start, end from kernels order by start").fetchall()
        seq = [(short(n), (e - st) / 1e3) for n, st, e in seq if re.search(a.calls, n)][-a.ncalls:]
        lines += ["", f"last {len(seq)} launches matching `{a.calls}`, in launch order (us)", "", "| kernel | us |", "|---|---:|"]
        lines += [f"| `{n}` | {d:.1f} |" for n, d in seq]
    txt = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    else:
        print(txt)


if __name__ == "__main__":
    main()
