#!/bin/bash
O=gpurun_out/r6j; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dist.py -q -x 2>&1 | tail -4) > $O/pytest_dist.log
grep -h "passed\|failed" $O/pytest_dist.log
for i in 1 2; do
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 --optimizer adagrad --rows-per-gpu 125000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('adagrad 125m', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['loss'])"
done
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 --optimizer sgd --rows-per-gpu 125000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sgd 125m', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['loss'])"
