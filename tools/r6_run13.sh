#!/bin/bash
O=gpurun_out/r6m; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_fuzz.py tests/test_gpu_models.py -q -x 2>&1 | tail -14) > $O/pytest_emb.log
grep -h "passed\|failed" $O/pytest_emb.log
for i in 1 2; do
for v in "--zipf --field-layout skewed" "--zipf" "--field-layout skewed" ""; do
timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['ms_per_step'], d['config']['loss'])"
done; done
TRS_TIMELINE=$O/skewed_zipf_timeline.md timeout 600 bash tools/trace_run.sh $O/skewed_zipf_trace.md "r06 -- DeepFM, criteo-skewed field sizes + Zipf(1.05) indices" -- python $PWD/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 20 --warmup 5 --zipf --field-layout skewed
grep "scatter_long" $O/skewed_zipf_timeline.md
