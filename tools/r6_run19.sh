#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_cin_parity.py tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_layers.py -q 2>&1 | tail -12 | grep -h "passed\|failed"
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-large-table --model xdeepfm --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm', d['ms_per_step'], d['config']['loss'])"
done
