#!/bin/bash
# usage (on the GPU box): tools/pmc_traffic.sh <out-dir> <kernel-name-filter> -- <command ...>
# HBM traffic counters, one per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass), counters-only (--kernel-trace)
out=$1; filt=$2; shift 3
mkdir -p "$out"
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o t -- "$@" > /tmp/pmc_$c.log 2>&1
done
cd "$R"
for c in FETCH_SIZE WRITE_SIZE; do
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/pmc_summary.py "$db" $filt > "$out/pmc_$c.txt" 2>&1; else tail -5 /tmp/pmc_$c.log > "$out/pmc_$c.txt"; fi
done
cat "$out"/pmc_FETCH_SIZE.txt "$out"/pmc_WRITE_SIZE.txt
