#!/bin/bash
for v in one two inputs two_big; do
  s=$(date +%s); TRS_SHARD_FORCE_COLLECTIVES=1 timeout 120 python tools/experiments/rccl_graph_probe.py $v 2>/dev/null | tail -3 | tr '\n' ' '; echo " [$v: $(( $(date +%s) - s )) s]"
done
