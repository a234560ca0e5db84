#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/quick_gpu.sh <tag> [pytest args...]
# one development iteration: a pytest subset, the DeepFM bench line (short form), its kernel trace and one step's timeline
tag=${1:-dev}; shift
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p "$O"
if [ $# -gt 0 ]; then (timeout 1200 python -m pytest "$@" 2>&1 | tail -25) > $O/pytest.log; tail -4 $O/pytest.log; fi
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models 2>$O/bench.err | tail -1 > $O/bench_deepfm_$i.json; python - <<PY
import json; d=json.loads(open("$O/bench_deepfm_$i.json").read() or "{}"); print("deepfm", d.get("ms_per_step"), d.get("config",{}).get("loss"), d.get("roofline",{}).get("avg_launch_us"))
PY
done
TRS_TIMELINE=$O/step_timeline.md timeout 300 bash tools/trace_run.sh $O/bench_deepfm_kernel_trace.md "$tag -- rocprofv3 --kernel-trace --stats: bench.py (DeepFM)" -- python $R/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 20 --warmup 5
tail -3 $O/step_timeline.md
