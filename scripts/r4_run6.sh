#!/bin/bash
# round-4 validation: full GPU suite on the default build, 16-bit-ring tests + fuzz + stress under the bounds-checked build, C5 A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== full GPU suite (default build)"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t6.log 2>&1; rc=$?; tail -4 gpurun_out/r04_t6.log
grep -q "Memory access fault" gpurun_out/r04_t6.log && { echo "FAULT in the suite"; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t6.log | head -5; exit 1; }
echo "== 16-bit ring / 50 kb / several pairs under the bounds-checked build"
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "16bit_ring or c5_three or several_pairs or 50kb or deep_levels" > gpurun_out/r04_t6b.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t6b.log
grep -q "Memory access fault" gpurun_out/r04_t6b.log && { echo "FAULT under bounds"; exit 1; }
[ $rc -ne 0 ] && exit 1
rm -f gpurun_out/r04_ab_c5.log; AB_STEPS="--steps 3 --warmup 1" bash scripts/r4_ab.sh r04_ab_c5.log "default nopku16" 2 --config C5 --nseq 32
echo "== fuzz + stress under the bounds-checked build"
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 420 python scripts/gpu_fuzz.py 12 120 40 > gpurun_out/r04_fuzz_bounds.log 2>&1; echo "fuzz rc=$?"; grep -n "failures\|FAIL\|fault" gpurun_out/r04_fuzz_bounds.log | head
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 300 python scripts/stress_multi.py 20 > gpurun_out/r04_stress_bounds.log 2>&1; echo "stress rc=$?"; tail -2 gpurun_out/r04_stress_bounds.log
