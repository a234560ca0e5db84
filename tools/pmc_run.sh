#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <out-dir> <kernel-name-filter> -- <command ...>
# two counters-only passes of rocprofv3 (--pmc with --kernel-trace only), summarised per kernel by tools/pmc_summary.py
out=$1; filt=$2; shift 3
mkdir -p "$out"
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rm -rf /tmp/pmc_a /tmp/pmc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_a -o a -- "$@" > /tmp/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_b -o b -- "$@" > /tmp/pmc_b.log 2>&1
cd "$R"
for p in a b; do
  db=$(find /tmp/pmc_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/pmc_summary.py "$db" $filt > "$out/pmc_$p.txt" 2>&1; else tail -5 /tmp/pmc_$p.log > "$out/pmc_$p.txt"; fi
done
cat "$out"/pmc_a.txt "$out"/pmc_b.txt
