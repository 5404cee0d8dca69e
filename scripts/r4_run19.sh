#!/bin/bash
# round 4, run 19: the 32-bit searches forced onto short sequences (SR_FORCE_INT32), bounds-checked build first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== bounds build"; SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "32bit_searches_on_short" > gpurun_out/r04_t19a.log 2>&1; rc=$?; tail -15 gpurun_out/r04_t19a.log | cut -c1-400
grep -q "Memory access fault" gpurun_out/r04_t19a.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && exit 1
echo "== default build"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "32bit_searches_on_short" > gpurun_out/r04_t19b.log 2>&1; rc=$?; tail -15 gpurun_out/r04_t19b.log | cut -c1-400
