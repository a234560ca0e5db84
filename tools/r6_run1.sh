#!/bin/bash
# round 6, first GPU call: new tests, full GPU suite, default bench line, sharded one-rank lines (1 M and 125 M rows)
O=gpurun_out/r6a; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "step_full_size" 2>&1 | tail -40) > $O/pytest_steps.log
(timeout 600 python -m pytest tests/test_gpu_dist.py -q -x -k "cfg5" 2>&1 | tail -15) > $O/pytest_cfg5.log
(TRS_TOL_REPORT=$PWD/$O/tol_layers.tsv timeout 600 python -m pytest tests/test_gpu_layers.py -q 2>&1 | tail -5) > $O/pytest_layers.log
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/pytest_all.log
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench_default.json
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 2>$O/shard1m.err | tail -1 > $O/bench_shard_1m.json
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 --rows-per-gpu 125000000 2>$O/shard125m.err | tail -1 > $O/bench_shard_125m.json
timeout 600 python bench.py --no-cpu-baseline --no-other-models --no-large-table --steps 40 --rows-per-gpu 125000000 2>$O/single125m.err | tail -1 > $O/bench_single_125m.json
tail -3 $O/pytest_steps.log $O/pytest_cfg5.log $O/pytest_all.log
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read()); print("$f", d["ms_per_step"], d.get("config",{}).get("loss"), (d.get("roofline") or {}).get("frac"))
except Exception as e: print("$f", "FAILED", e)
PY
done
