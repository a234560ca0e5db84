#!/bin/bash
O=gpurun_out/r6g; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_fullsize.py -q -x -k "cross or dcn or layers" 2>&1 | tail -5) > $O/pytest_cross.log
grep -h "passed\|failed" $O/pytest_cross.log
for i in 1 2; do python tools/kbench.py --what cross 2>&1 | grep cross_bwd; done > $O/kbench_flags.txt
touch torecsys_amd/csrc/cross_mfma.hip
TRS_BUILD_DEFS="cross_mfma.hip:-DTRS_B3_BARRIER" python -m torecsys_amd.build > /dev/null 2>&1
for i in 1 2; do python tools/kbench.py --what cross 2>&1 | grep cross_bwd; done > $O/kbench_barrier.txt
touch torecsys_amd/csrc/cross_mfma.hip
python -m torecsys_amd.build > /dev/null 2>&1
for i in 1 2; do python tools/kbench.py --what cross 2>&1 | grep cross_bwd; done >> $O/kbench_flags.txt
echo flags; cat $O/kbench_flags.txt; echo barrier; cat $O/kbench_barrier.txt
timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded whole 1m', d['ms_per_step'], d['config']['loss'])"
timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps 40 --rows-per-gpu 125000000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded whole 125m', d['ms_per_step'], d['config']['loss'])"
