#!/bin/bash
O=gpurun_out/r6l; mkdir -p $O
TRS_TIMELINE=$O/skewed_zipf_timeline.md timeout 600 bash tools/trace_run.sh $O/skewed_zipf_trace.md "r06 -- DeepFM, criteo-skewed field sizes + Zipf(1.05) indices" -- python $PWD/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 20 --warmup 5 --zipf --field-layout skewed
TRS_TIMELINE=$O/zipf_timeline.md timeout 600 bash tools/trace_run.sh $O/zipf_trace.md "r06 -- DeepFM, Zipf(1.05) indices" -- python $PWD/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 20 --warmup 5 --zipf
