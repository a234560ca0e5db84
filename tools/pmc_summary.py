#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc results (sqlite .db) per kernel: mean counter value per dispatch.
usage: tools/pmc_summary.py <results.db> [name-substring ...]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
filters = sys.argv[2:]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = next((t for t in tabs if t == "counters_collection"), None)
if view is None:
    print("tables/views:", tabs)
    sys.exit("no counters_collection view in this file")
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
kcol = "kernel_name" if "kernel_name" in cols else next(c for c in cols if "kernel" in c and "name" in c)
rows = db.execute(f"select {kcol}, counter_name, value, dispatch_id from {view}").fetchall()
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for k, c, v, d in rows:
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    if filters and not any(f in k for f in filters):
        continue
    acc[k][c] += float(v)
    disp[k].add(d)
for k in sorted(acc):
    n = max(1, len(disp[k]))
    print(f"{k}  (dispatches {n})")
    for c in sorted(acc[k]):
        print(f"    {c:32s} {acc[k][c] / n:16.1f}")
