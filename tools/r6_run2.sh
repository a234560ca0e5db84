#!/bin/bash
# round 6, second GPU call: changed tests, lazy-zero A/B, timelines of the one-rank sharded step (1 M and 125 M rows)
O=gpurun_out/r6b; mkdir -p $O
(TRS_TOL_REPORT=$PWD/$O/tol_steps.tsv timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "step_full_size" 2>&1 | tail -8) > $O/pytest_steps.log
(timeout 1200 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_fuzz.py tests/test_gpu_graph.py tests/test_gpu_head.py tests/test_gpu_mlp.py tests/test_gpu_layers.py -q -x 2>&1 | tail -8) > $O/pytest_sub.log
bash tools/ab_env.sh r6b "TRS_CSR_LAZY_ZERO=0" "TRS_CSR_LAZY_ZERO=1"
TRS_TIMELINE=$O/shard_125m_timeline.md TRS_TIMELINE_ANCHOR=embed_fm timeout 600 bash tools/trace_run.sh $O/shard_125m_trace.md "r06 one-rank sharded step, 125 M rows" -- python $PWD/bench.py --force-sharded --no-cpu-baseline --steps 20 --warmup 5 --rows-per-gpu 125000000
TRS_TIMELINE=$O/shard_1m_timeline.md TRS_TIMELINE_ANCHOR=embed_fm timeout 600 bash tools/trace_run.sh $O/shard_1m_trace.md "r06 one-rank sharded step, 1 M rows" -- python $PWD/bench.py --force-sharded --no-cpu-baseline --steps 20 --warmup 5
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $O/pytest_all.log
tail -n 4 $O/pytest_steps.log $O/pytest_sub.log $O/pytest_all.log
cat $O/ab.txt
