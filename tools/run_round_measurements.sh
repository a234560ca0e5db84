#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/run_round_measurements.sh r04
# writes gpurun_out/<tag>/*: the bench lines of the three models and their variants, stand-alone kernel timings and the
# rocprofv3 kernel traces that profiles/<tag>_* are copied from
tag=${1:-r02}
R=$(pwd)
O=$R/gpurun_out/$tag
mkdir -p "$O"
run() { timeout 600 "$@"; }
run python bench.py > $O/bench_deepfm.json 2> $O/bench_deepfm.err; tail -1 $O/bench_deepfm.json | cut -c1-300
for m in fm dcn xdeepfm; do run python bench.py --no-cpu-baseline --no-large-table --model $m 2>/dev/null | tail -1 > $O/bench_$m.json; cut -c1-200 $O/bench_$m.json; done
run python bench.py --no-cpu-baseline --no-large-table --no-other-models --eager 2>/dev/null | tail -1 > $O/bench_deepfm_eager.json
run python bench.py --no-cpu-baseline --no-large-table --no-other-models --zipf 2>/dev/null | tail -1 > $O/bench_deepfm_zipf.json
run python bench.py --no-cpu-baseline --no-large-table --no-other-models --field-layout skewed 2>/dev/null | tail -1 > $O/bench_deepfm_skewed.json
run python bench.py --no-cpu-baseline --no-large-table --no-other-models --field-layout skewed --zipf 2>/dev/null | tail -1 > $O/bench_deepfm_skewed_zipf.json
run python bench.py --no-cpu-baseline --no-large-table --no-other-models --optimizer adagrad 2>/dev/null | tail -1 > $O/bench_deepfm_adagrad.json
run python bench.py --no-cpu-baseline --no-large-table --force-sharded 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1.json
run python bench.py --no-cpu-baseline --no-large-table --force-sharded --rows-per-gpu 125000000 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_125m.json
run python bench.py --no-cpu-baseline --no-large-table --force-sharded --shard-graph region 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_region.json
run python bench.py --no-cpu-baseline --no-large-table --force-sharded --shard-graph region --rows-per-gpu 125000000 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_region_125m.json
TRS_SHARD_LOCAL_DIRECT=0 run python bench.py --no-cpu-baseline --no-large-table --force-sharded 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_buffers.json
TRS_SHARD_LOCAL_DIRECT=0 run python bench.py --no-cpu-baseline --no-large-table --force-sharded --rows-per-gpu 125000000 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_buffers_125m.json
run python bench.py --no-cpu-baseline --no-large-table --force-sharded --optimizer adagrad 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_adagrad.json
run python bench.py --no-cpu-baseline --no-large-table --force-sharded --optimizer adagrad --rows-per-gpu 125000000 2>/dev/null | tail -1 > $O/bench_deepfm_sharded1_adagrad_125m.json
run python tools/kbench.py 2>&1 | grep -v Warn > $O/kbench.txt
run python tools/kbench.py --what pairx,mlpf 2>&1 | grep -v Warn > $O/kbench_pairx_mlpf.txt
run python tools/kbench.py --what ffm 2>&1 | grep -v Warn > $O/kbench_ffm.txt
# HBM traffic of the roofline kernel from the PMC counters (separate passes, counters only) -> profiles/traffic.json
run bash tools/pmc_traffic.sh $O/pmc_traffic "embed_fm_group scatter_rows_fm1" -- python $R/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 10 --warmup 3 --eager > /dev/null 2>&1
python tools/make_traffic_json.py $O/pmc_traffic ${tag#r} --md $O/pmc_embed_fm.md; cp profiles/traffic.json $O/traffic.json
TRS_TIMELINE=$O/bench_deepfm_step_timeline.md run bash tools/trace_run.sh $O/bench_deepfm_kernel_trace.md "$tag -- rocprofv3 --kernel-trace --stats: bench.py (DeepFM, BASELINE configs[1])" -- python $R/bench.py --no-cpu-baseline --no-large-table --no-other-models --steps 20 --warmup 5
TRS_TRACE_CALLS="cin_|glue" run bash tools/trace_run.sh $O/bench_xdeepfm_kernel_trace.md "$tag -- rocprofv3 --kernel-trace --stats: bench.py --model xdeepfm" -- python $R/bench.py --no-cpu-baseline --no-large-table --model xdeepfm --steps 3 --warmup 2
run bash tools/trace_run.sh $O/bench_dcn_kernel_trace.md "$tag -- rocprofv3 --kernel-trace --stats: bench.py --model dcn" -- python $R/bench.py --no-cpu-baseline --no-large-table --model dcn --steps 3 --warmup 2
ls -la $O
