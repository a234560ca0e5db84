#!/bin/bash
# usage (GPU box): bash tools/ab_env.sh <tag> "VAR=a VAR2=b" "VAR=c" ...   -- the DeepFM bench line under each environment,
# alternately, three rounds (same box: box-to-box differences are larger than most of what is being compared)
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
for r in 1 2 3; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    ms=$(env $e timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])")
    echo "round $r  [$e]  $ms" | tee -a $O/ab.txt
  done
done
