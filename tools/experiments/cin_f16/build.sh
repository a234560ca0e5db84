#!/bin/bash
# Builds tools/experiments/cin_f16/libcin16.so (the packed-fp16 CIN kernels of round 4, NOT part of libtrs_hip.so).
#   bash tools/experiments/cin_f16/build.sh [extra hipcc flags, e.g. -DTRS_CIN16_ABL=7 -DTRS_CIN16_STAGE_REG] [-o name]
set -e
here=$(cd $(dirname $0) && pwd); root=$(cd $here/../../.. && pwd)
out=libcin16.so
args=()
while [ $# -gt 0 ]; do if [ "$1" == "-o" ]; then out=$2; shift 2; else args+=("$1"); shift; fi; done
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I$root/include \
    "${args[@]}" -c $here/cin_f16.hip -o $here/${out%.so}.o
g++ -shared -o $here/$out $here/${out%.so}.o $root/torecsys_amd/libtrs_hip.so -L$TL -lamdhip64 -Wl,-rpath,$root/torecsys_amd
rm -f $here/${out%.so}.o
echo built $here/$out
