#!/bin/bash
O=gpurun_out/r6z; mkdir -p $O
s=$(date +%s); TRS_BENCH_STAGES=1 timeout 900 python bench.py 2>$O/err.txt | tail -1 > $O/bench_default.json; grep "bench stage" $O/err.txt | grep -v "warm\|timed region done\|capture\|index ring\|modules built"; echo "total $(( $(date +%s) - s )) s"
python - <<PY
import json; d=json.loads(open("$O/bench_default.json").read()); print(d["ms_per_step"], d["roofline"]["frac"], {k:v["ms_per_step"] for k,v in d["other_models"].items()}, {k:v.get("ms_per_step") for k,v in d["variants"].items()}, d["cpu_baseline"]["value"])
PY
s=$(date +%s); timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['other_models'].items()}, {k:v.get('ms_per_step') for k,v in d['variants'].items()}, d['cpu_baseline']['value'], d['self_check']['ok'], [v['self_check']['ok'] for v in d['other_models'].values()])"; echo "second total $(( $(date +%s) - s )) s"
