#!/bin/bash
# usage (on the GPU box): tools/trace_run.sh <out.md> "<title>" -- <command ...>     (rocprofv3 --kernel-trace --stats)
out=$1; title=$2; shift 3
export TMPDIR=/tmp
R=$(pwd)
mkdir -p "$(dirname "$out")"
cd /tmp && rm -rf /tmp/trc && rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- "$@" > /tmp/trc.log 2>&1
cd "$R"
db=$(find /tmp/trc -name "*.db" | head -1)
python tools/prof_summary.py "$db" --out "$out" --title "$title" --cmd "rocprofv3 --kernel-trace --stats -- $*" ${TRS_TRACE_CALLS:+--calls "$TRS_TRACE_CALLS"}
# TRS_TIMELINE=<file.md> [TRS_TIMELINE_ANCHOR=<kernel substring>]: one step of the same trace, launch by launch
if [ -n "$TRS_TIMELINE" ]; then python tools/step_timeline.py "$db" --out "$TRS_TIMELINE" --title "$title -- one step" ${TRS_TIMELINE_ANCHOR:+--anchor "$TRS_TIMELINE_ANCHOR"}; fi
