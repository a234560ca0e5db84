#!/bin/bash
O=gpurun_out/s5a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models 2>/dev/null | tail -1 > $O/bench_deepfm_$i.json; python -c "import json;d=json.load(open('$O/bench_deepfm_$i.json'));print(d['ms_per_step'],d['roofline']['frac'],d['config'].get('loss'))"; done
TRS_TIMELINE=$O/timeline_marked.md timeout 600 bash tools/trace_run.sh $O/trace.md "s5a" -- python $(pwd)/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 20 --warmup 5
db=$(find /tmp/trc -name "*.db" | head -1)
python tools/step_timeline.py "$db" --out $O/timeline_timed.md --title "timed replay step" --step -14
grep -n "bce" $O/trace.md
