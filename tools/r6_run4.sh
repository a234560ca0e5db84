#!/bin/bash
O=gpurun_out/r6d; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_graph.py tests/test_gpu_embedding.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -8) > $O/pytest_dist.log
tail -n 3 $O/pytest_dist.log
run() { name=$1; shift; timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 "$@" 2>$O/$name.err | tail -1 > $O/$name.json
python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read()); c=d["config"]; print("$name", d["ms_per_step"], "host", c.get("host_enqueue_ms_per_step"), "loss", c.get("loss"), c.get("hipgraph_scope","")[:12], json.dumps((c.get("sharded") or {}).get("phase_ms_per_call")))
except Exception as e: print("$name", "FAILED", e); import subprocess; print(subprocess.run(["tail","-5","$O/$name.err"],capture_output=True,text=True).stdout)
PY
}
run whole_1m
run whole_125m --rows-per-gpu 125000000
run region_1m --shard-graph region
TRS_SHARD_OWN_DIRECT=0 run whole_1m_ownperm
TRS_SHARD_LOCAL_DIRECT=0 run whole_1m_buffers
TRS_SHARD_LOCAL_DIRECT=0 run whole_125m_buffers --rows-per-gpu 125000000
run whole_1m_zipf --zipf
run whole_1m_sgd --optimizer sgd
timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unsharded', d['ms_per_step'])"
TRS_TIMELINE=$O/whole_1m_timeline.md TRS_TIMELINE_ANCHOR=embed_fm_sharded timeout 600 bash tools/trace_run.sh $O/whole_1m_trace.md "r06 one-rank sharded step (whole-step graph, local-direct), 1 M rows" -- python $PWD/bench.py --force-sharded --no-cpu-baseline --steps 20 --warmup 5
