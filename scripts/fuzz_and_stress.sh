#!/bin/bash
# round-3 end state: randomised parity (small / medium / mixed sets) and the multi-pair stress against the level-per-pass kernel
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python scripts/gpu_fuzz.py 12 260 80 > gpurun_out/r03_gpu_fuzz.log 2>&1; echo "fuzz rc=$?"; grep -n "failures\|FAIL" gpurun_out/r03_gpu_fuzz.log | head -20
timeout -k 10 600 python scripts/stress_multi.py 40 > gpurun_out/r03_stress_multi.log 2>&1; echo "stress rc=$?"; tail -3 gpurun_out/r03_stress_multi.log
