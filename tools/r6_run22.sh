#!/bin/bash
O=gpurun_out/r6v; mkdir -p $O
for v in 0 1; do
s=$(date +%s)
TRS_SHARD_FORCE_COLLECTIVES=$v TRS_SHARD_LOCAL_DIRECT=0 timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps 40 --rows-per-gpu 125000000 2>$O/fc$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('buffers 125m whole-step graph, force_collectives=$v', d['ms_per_step'], c['loss'], c.get('hipgraph_scope','')[:10], 'host', c['host_enqueue_ms_per_step'])"
echo "   took $(( $(date +%s) - s )) s"
done
TRS_SHARD_FORCE_COLLECTIVES=1 TRS_SHARD_LOCAL_DIRECT=0 timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps 40 --rows-per-gpu 125000000 --shard-graph region 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('buffers 125m region, force_collectives=1', d['ms_per_step'], c['loss'], 'host', c['host_enqueue_ms_per_step'])"
