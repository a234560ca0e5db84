#!/bin/bash
s=$(date +%s); TRS_BENCH_STAGES=1 timeout 900 python bench.py 2>&1 >/dev/null | grep "bench stage"; echo "total $(( $(date +%s) - s )) s"
s=$(date +%s); TRS_BENCH_STAGES=1 timeout 900 python bench.py 2>&1 >/dev/null | grep "bench stage"; echo "second run total $(( $(date +%s) - s )) s"
