set -x
cd /root/repo
mkdir -p gpurun_out/r01j
python bench.py > gpurun_out/r01j/bench_deepfm.json 2> gpurun_out/r01j/bench_deepfm.err; tail -1 gpurun_out/r01j/bench_deepfm.json | cut -c1-400
for m in fm dcn xdeepfm; do python bench.py --no-cpu-baseline --model $m --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r01j/bench_$m.json; cut -c1-200 gpurun_out/r01j/bench_$m.json; done
python bench.py --no-cpu-baseline --graph 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_graph.json
python bench.py --no-cpu-baseline --graph --host-indices 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_graph_host.json
python bench.py --no-cpu-baseline --zipf 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_zipf.json
python bench.py --no-cpu-baseline --optimizer sgd 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_sgd.json
python bench.py --no-cpu-baseline --optimizer adam 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_adam.json
python bench.py --no-cpu-baseline --batch 16384 --graph 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_b16k_graph.json
python bench.py --no-cpu-baseline --force-sharded 2>/dev/null | tail -1 > gpurun_out/r01j/bench_deepfm_sharded1.json
python tools/kbench.py 2>&1 | grep -v Warn > gpurun_out/r01j/kbench.txt
python tools/kbench.py --what pairx 2>&1 | grep -v Warn > gpurun_out/r01j/kbench_pairx.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o b -- python /root/repo/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /tmp/bench_prof.out 2>&1
tail -1 /tmp/bench_prof.out > /root/repo/gpurun_out/r01j/bench_under_rocprof.json
python /root/repo/tools/prof_summary.py $(find /tmp/p_bench -name "*.db" | head -1) --out /root/repo/gpurun_out/r01j/bench_kernel_trace.md --title "rocprofv3 --kernel-trace --stats: python bench.py --steps 20 --warmup 5 --no-cpu-baseline" --cmd "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 20 --warmup 5"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_x -o b -- python /root/repo/bench.py --no-cpu-baseline --model xdeepfm --steps 3 --warmup 2 > /dev/null 2>&1
python /root/repo/tools/prof_summary.py $(find /tmp/p_x -name "*.db" | head -1) --out /root/repo/gpurun_out/r01j/bench_xdeepfm_kernel_trace.md --title "rocprofv3 --kernel-trace --stats: bench.py --model xdeepfm" --cmd "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --model xdeepfm --steps 3 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_d -o b -- python /root/repo/bench.py --no-cpu-baseline --model dcn --steps 3 --warmup 2 > /dev/null 2>&1
python /root/repo/tools/prof_summary.py $(find /tmp/p_d -name "*.db" | head -1) --out /root/repo/gpurun_out/r01j/bench_dcn_kernel_trace.md --title "rocprofv3 --kernel-trace --stats: bench.py --model dcn" --cmd "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --model dcn --steps 3 --warmup 2"
ls -la /root/repo/gpurun_out/r01j
