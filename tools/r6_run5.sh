#!/bin/bash
O=gpurun_out/r6e; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_dist.py -q -x 2>&1 | tail -4) > $O/pytest_dist.log
tail -n 2 $O/pytest_dist.log
run() { name=$1; shift; timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 40 "$@" 2>$O/$name.err | tail -1 > $O/$name.json
python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read()); c=d["config"]; print("$name", d["ms_per_step"], "host", c.get("host_enqueue_ms_per_step"), "loss", c.get("loss"))
except Exception as e: print("$name", "FAILED", e)
PY
}
run whole_1m
run whole_125m --rows-per-gpu 125000000
python tools/kbench.py --what cross > $O/kbench_cross_default.txt 2>&1
bash tools/pmc_run.sh $O/pmc_cross cross_mfma -- python $PWD/tools/kbench.py --what cross > /dev/null 2>&1
bash tools/pmc_run.sh $O/pmc_cin cin_ -- python $PWD/tools/kbench.py --what cin > /dev/null 2>&1
python tools/pmc_table.py $O/pmc_cross $O/pmc_cin > $O/pmc_table.md 2>&1
python tools/kbench.py --what cin > $O/kbench_cin.txt 2>&1
# A/B: cross_mfma.hip compiled WITH the SLP vectoriser (packed fp32 math in the VALU-bound chain waves)
touch torecsys_amd/csrc/cross_mfma.hip
TRS_BUILD_SLP=cross_mfma.hip python -m torecsys_amd.build > $O/build_slp.log 2>&1
python tools/kbench.py --what cross > $O/kbench_cross_slp.txt 2>&1
touch torecsys_amd/csrc/cross_mfma.hip
python -m torecsys_amd.build >> $O/build_slp.log 2>&1
grep -h "cross" $O/kbench_cross_default.txt $O/kbench_cross_slp.txt
cat $O/pmc_table.md
cat $O/kbench_cin.txt | tail -12
