#!/bin/bash
O=gpurun_out/r6i; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dist.py -q -x 2>&1 | tail -4) > $O/pytest_dist.log
grep -h "passed\|failed" $O/pytest_dist.log
run() { name=$1; shift; timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 "$@" 2>$O/$name.err | tail -1 > $O/$name.json
python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read()); c=d["config"]; print("$name", d["ms_per_step"], "host", c.get("host_enqueue_ms_per_step"), "loss", c.get("loss"), json.dumps({k:v for k,v in (c.get("sharded") or {}).get("phase_ms_per_call",{}).items() if "owner reduce" in k}))
except Exception as e: print("$name", "FAILED", e)
PY
}
run adagrad_125m --optimizer adagrad --rows-per-gpu 125000000
TRS_SHARD_DENSE_INDEX_ROWS=0 run adagrad_125m_compact --optimizer adagrad --rows-per-gpu 125000000
run adagrad_125m_b --optimizer adagrad --rows-per-gpu 125000000
run sgd_125m --optimizer sgd --rows-per-gpu 125000000
bash tools/pmc_run.sh $O/pmc_pairx "pair" -- python $PWD/tools/kbench.py --what pairx > /dev/null 2>&1
bash tools/pmc_run.sh $O/pmc_afm "afm" -- python $PWD/tools/kbench.py --what pairx > /dev/null 2>&1
python tools/pmc_table.py $O/pmc_pairx $O/pmc_afm > $O/pmc_table_pairx.md 2>&1
cat $O/pmc_table_pairx.md
