#!/bin/bash
O=gpurun_out/r6p; mkdir -p $O
for v in 0 1; do
TRS_SHARD_FORCE_COLLECTIVES=$v TRS_SHARD_LOCAL_DIRECT=0 timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 --rows-per-gpu 125000000 2>$O/fc$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('buffers 125m force_collectives=$v', d['ms_per_step'], c['loss'], c.get('hipgraph_scope','')[:10], c['host_enqueue_ms_per_step'])"
done
tail -2 $O/fc1.err
