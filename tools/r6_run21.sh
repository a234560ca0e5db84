#!/bin/bash
O=gpurun_out/r6u; mkdir -p $O
TRS_BENCH_WATCHDOG=100 TRS_SHARD_FORCE_COLLECTIVES=1 TRS_SHARD_LOCAL_DIRECT=0 timeout 160 python bench.py --force-sharded --no-cpu-baseline --steps 8 --warmup 2 --batch 8192 --graph-steps-per-replay 1 > $O/out.txt 2> $O/err.txt
grep -v "Warning\|warn\|amdgpu.ids\|names = " $O/err.txt | tail -60
