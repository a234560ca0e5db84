#!/bin/bash
O=gpurun_out/r6h; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_cin_parity.py tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_layers.py -q -x 2>&1 | tail -6) > $O/pytest_cin.log
grep -h "passed\|failed" $O/pytest_cin.log
for v in 1 0 1 0; do
TRS_CIN_SKIP_DEAD=$v timeout 600 python bench.py --no-cpu-baseline --no-large-table --model xdeepfm --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xdeepfm skip_dead=$v', d['ms_per_step'], d['config']['loss'], (d.get('roofline_model_kernel') or {}).get('frac'))"
done
