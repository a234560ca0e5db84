#!/bin/bash
bash tools/ab_env_args.sh s5h "" "TRS_CSR_TARGET=0" "TRS_CSR_TARGET=128" "TRS_CSR_TARGET=192" "TRS_CSR_TARGET=320" 2>&1 | tail -12
for t in 0 128 192 320; do TRS_CSR_TARGET=$t timeout 300 python tools/kbench.py --what csr 2>&1 | grep csr_build; done
