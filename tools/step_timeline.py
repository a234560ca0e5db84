#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace run as a timeline: every launch between two launches of an anchor kernel, with
its start offset, duration, queue and the idle gap in front of it on its own queue -- where the step's dependency bubbles
and its side-stream overlap are.
usage: tools/step_timeline.py <results.db> [--anchor embed_fm_group] [--step -2] [--out file.md] [--title "..."]"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if name.startswith("Cijk_") or name.startswith("Custom_Cijk"):
        m = re.search(r"(MT\d+x\d+x\d+)", name)
        return "hipBLASLt " + (m.group(1) if m else name[:20])
    name = re.sub(r"<.*", "", name)
    return name if len(name) <= 60 else name[:57] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--anchor", default="embed_fm_group")
    ap.add_argument("--step", type=int, default=-2, help="which anchor-to-anchor interval (negative: from the end)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--title", default="one step, launch by launch")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((q for q in ("stream_id", "queue_id", "stream", "queue") if q in cols), None)
    sel = f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"
    rows = c.execute(sel).fetchall()
    anchors = [i for i, r in enumerate(rows) if a.anchor in r[0]]
    if len(anchors) < 2:
        raise SystemExit(f"fewer than two launches of {a.anchor!r}; columns of `kernels`: {cols}")
    k = a.step if a.step >= 0 else len(anchors) - 1 + a.step
    k = max(0, min(k, len(anchors) - 2))
    i0, i1 = anchors[k], anchors[k + 1]
    # launches of side queues that START inside the interval belong to the step as well (rows are ordered by start)
    step = rows[i0:i1]
    t0 = step[0][1]
    span = (rows[i1][1] - t0) / 1e3
    last_end = {}
    busy = {}
    lines = [f"# {a.title}", "", f"anchor `{a.anchor}`, interval {k} of {len(anchors) - 1}: {span:.1f} us from its launch to the next one's; "
             f"{len(step)} launches (queue column: {qcol})", "",
             "| start (us) | dur (us) | gap on its queue (us) | queue | kernel |", "|---:|---:|---:|---:|---|"]
    for r in step:
        name, st, en = r[0], r[1], r[2]
        q = r[3] if qcol else 0
        gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(en, last_end.get(q, 0))
        busy[q] = busy.get(q, 0.0) + (en - st) / 1e3
        lines.append(f"| {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {gap:.1f} | {q} | `{short(name)}` |")
    lines += ["", "busy time per queue (us): " + ", ".join(f"{q}: {b:.1f}" for q, b in sorted(busy.items(), key=lambda x: -x[1]))]
    # union of all busy intervals = time the device ran at least one kernel
    iv = sorted((r[1], r[2]) for r in step)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    tot += ce - cs
    lines += [f"device busy (union over queues): {tot / 1e3:.1f} us of {span:.1f} us"]
    txt = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    else:
        print(txt)


if __name__ == "__main__":
    main()
