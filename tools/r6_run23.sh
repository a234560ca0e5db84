#!/bin/bash
O=gpurun_out/r6w; mkdir -p $O
s=$(date +%s); timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json; echo "default bench took $(( $(date +%s) - s )) s"
python - <<PY
import json; d=json.loads(open("$O/bench_default.json").read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_round"], {k:v["ms_per_step"] for k,v in d["other_models"].items()}, {k:v.get("ms_per_step") for k,v in d["variants"].items()}, [v.get("self_check",{}).get("ok") for v in d["other_models"].values()], d["self_check"]["ok"])
PY
timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded 1m', d['ms_per_step'], d['config']['loss'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
