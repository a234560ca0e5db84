#!/bin/bash
# usage: pmc_ro.sh <tag> : HBM-side counters of the fused MLP forward (env decides which kernel)
export TMPDIR=/tmp
R=$(pwd)
mkdir -p $R/gpurun_out/pmc_ro
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_x
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_x -o x -- python $R/tools/mlp_ro_one.py > /tmp/pmc_x.log 2>&1
  db=$(find /tmp/pmc_x -name "*.db" | head -1)
  echo "== $1: $c" >> $R/gpurun_out/pmc_ro/$1.txt
  if [ -n "$db" ]; then python $R/tools/pmc_summary.py "$db" mlp_ >> $R/gpurun_out/pmc_ro/$1.txt 2>&1; else tail -3 /tmp/pmc_x.log >> $R/gpurun_out/pmc_ro/$1.txt; fi
done
cat $R/gpurun_out/pmc_ro/$1.txt
