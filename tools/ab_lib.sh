#!/bin/bash
# usage (GPU box): bash tools/ab_lib.sh <tag> <other libtrs .so> [bench.py args...]  -- the bench line with the in-tree
# library and with another build of it (TRS_LIB_PATH), alternately, three rounds on the same box
tag=$1; lib=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for r in 1 2 3; do
  for which in tree other; do
    if [ $which = other ]; then export TRS_LIB_PATH=$lib; else unset TRS_LIB_PATH; fi
    ms=$(timeout 300 python bench.py --no-cpu-baseline --no-large-table --no-other-models "$@" 2>>$O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'), d['config'].get('loss'))")
    echo "round $r  [$which] [$*]  $ms" | tee -a $O/ab.txt
  done
done
