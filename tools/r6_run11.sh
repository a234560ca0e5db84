#!/bin/bash
O=gpurun_out/r6k; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dist.py -q -x 2>&1 | tail -14) > $O/pytest_dist.log
grep -h "passed\|failed" $O/pytest_dist.log
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 --optimizer sgd --rows-per-gpu 125000000 2>$O/sgd.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sgd 125m', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['loss'])"
tail -3 $O/sgd.err
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 --optimizer sgd 2>$O/sgd1m.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sgd 1m', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config']['loss'], d['config'].get('hipgraph_scope'))"
