#!/bin/bash
# usage (GPU box): bash tools/ab_env_args.sh <tag> "<bench.py args>" "VAR=a" "VAR=b" ...   -- like ab_env.sh with extra
# bench.py arguments (e.g. "--force-sharded --no-pipeline"); prints ms_per_step and host enqueue per step
tag=$1; args=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for r in 1 2 3; do
  for e in "$@"; do
    ms=$(env $e timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models $args 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['config'].get('loss'))")
    echo "round $r  [$e] [$args]  $ms" | tee -a $O/ab.txt
  done
done
