#!/bin/bash
# end-of-round run on one box: what the driver runs (smoke, default bench line, full -m gpu suite), the suite's risky parts
# under the bounds-checked build, then the round's measurements (scripts/round_final.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/driver_like.sh 2>&1 | grep -v "amdgpu.ids" | cut -c1-500
echo "== full GPU suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite.log 2>&1; rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
grep -q "Memory access fault" gpurun_out/r04_gpu_suite.log && { echo "FAULT in the suite"; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_gpu_suite.log | head -5; exit 1; }
echo "== fuzz + stress under the bounds-checked build"
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 420 python scripts/gpu_fuzz.py 12 120 40 > gpurun_out/r04_fuzz_bounds.log 2>&1; echo "fuzz rc=$?"; grep -n "failures\|FAIL\|fault" gpurun_out/r04_fuzz_bounds.log | head
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 300 python scripts/stress_multi.py 20 > gpurun_out/r04_stress_bounds.log 2>&1; echo "stress rc=$?"; tail -1 gpurun_out/r04_stress_bounds.log
