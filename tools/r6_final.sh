#!/bin/bash
# round 6 final sweep: full GPU suite, every bench line / trace / counter profiles/r06_* are made from
O=gpurun_out/r06; mkdir -p $O
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_gpu_full.log
grep -h "passed\|failed" $O/pytest_gpu_full.log
bash tools/run_round_measurements.sh r06 > $O/sweep.log 2>&1
bash tools/pmc_run.sh $O/pmc_cross cross_mfma -- python $PWD/tools/kbench.py --what cross > /dev/null 2>&1
bash tools/pmc_run.sh $O/pmc_cin cin_ -- python $PWD/tools/kbench.py --what cin > /dev/null 2>&1
python tools/pmc_table.py $O/pmc_cross $O/pmc_cin > $O/pmc_table.md 2>&1
TRS_TIMELINE=$O/bench_sharded1_step_timeline.md TRS_TIMELINE_ANCHOR=embed_fm_sharded timeout 600 bash tools/trace_run.sh $O/bench_sharded1_kernel_trace.md "r06 -- rocprofv3 --kernel-trace --stats: bench.py --force-sharded (one rank, 1 M rows, whole-step graph)" -- python $PWD/bench.py --force-sharded --no-cpu-baseline --steps 20 --warmup 5
TRS_TIMELINE=$O/bench_sharded1_125m_step_timeline.md TRS_TIMELINE_ANCHOR=embed_fm_sharded timeout 600 bash tools/trace_run.sh $O/bench_sharded1_125m_kernel_trace.md "r06 -- rocprofv3 --kernel-trace --stats: bench.py --force-sharded --rows-per-gpu 125000000 (one rank, whole-step graph)" -- python $PWD/bench.py --force-sharded --no-cpu-baseline --steps 20 --warmup 5 --rows-per-gpu 125000000
tail -5 $O/sweep.log
