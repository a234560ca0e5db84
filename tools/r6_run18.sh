#!/bin/bash
O=gpurun_out/r6r; mkdir -p $O
t() { name=$1; shift; s=$(date +%s); TRS_SHARD_FORCE_COLLECTIVES=1 TRS_SHARD_LOCAL_DIRECT=0 timeout 150 python bench.py --force-sharded --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>$O/$name.err | tail -1 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['ms_per_step'], c['loss'], c.get('hipgraph_scope','')[:10])
except Exception as e: print('$name', 'NO RESULT')"; echo "   $name took $(( $(date +%s) - s )) s"; }
t b8192_spr1 --batch 8192 --graph-steps-per-replay 1
t b8192_spr4 --batch 8192
t b65536_spr1 --graph-steps-per-replay 1
t b65536_eager --eager
t b65536_region --shard-graph region
