#!/bin/bash
O=gpurun_out/r6f; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_dist.py -q -x -k "rccl_inside" 2>&1 | tail -6) > $O/pytest_rccl_graph.log
tail -n 3 $O/pytest_rccl_graph.log
(timeout 900 python -m pytest tests/test_gpu_dist.py -q -x 2>&1 | tail -4) > $O/pytest_dist.log
tail -n 2 $O/pytest_dist.log
bash tools/pmc_run.sh $O/pmc_cross cross_mfma -- python $PWD/tools/kbench.py --what cross > /dev/null 2>&1
bash tools/pmc_run.sh $O/pmc_cin cin_ -- python $PWD/tools/kbench.py --what cin > /dev/null 2>&1
python tools/pmc_table.py $O/pmc_cross $O/pmc_cin > $O/pmc_table.md 2>&1
cat $O/pmc_table.md
