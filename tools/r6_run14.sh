#!/bin/bash
O=gpurun_out/r6n; mkdir -p $O
TRS_ROWS_REPORT=$PWD/$O/rows.tsv timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | grep -h "passed\|failed"
python - <<PY
import collections
d=collections.defaultdict(lambda:[0,0.0,0.0])
for ln in open("$O/rows.tsv"):
    k,_,r,_,g=ln.strip().split("\t"); r=float(r); g=float(g)
    e=d[k]; e[0]+=1; e[1]=max(e[1],r); e[2]=max(e[2],g)
rows=sorted(d.items(), key=lambda kv:-kv[1][1]/max(kv[1][2],1e-30))
print(len(rows),"assert sites")
for k,(n,r,g) in rows[:60]: print(f"{k:40s} n={n:4d} rows_max={r:.2e} global_max={g:.2e} ratio={r/max(g,1e-30):.1f}")
PY
