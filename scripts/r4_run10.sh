#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== parity subset (default)"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or several_pairs or requeue or randomised or scaled_baseline or parity_cases or 16bit_ring or c5_three or orientation or deep_levels or kernels_and_workgroup" > gpurun_out/r04_t10.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t10.log
grep -q "Memory access fault" gpurun_out/r04_t10.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t10.log | head -5; exit 1; }
rm -f gpurun_out/r04_ab_band.log; bash scripts/r4_ab.sh r04_ab_band.log "default noband" 2
rm -f gpurun_out/r04_ab_band_c3.log; bash scripts/r4_ab.sh r04_ab_band_c3.log "default noband" 2 --config C3
rm -f gpurun_out/r04_ab_band_c4.log; AB_STEPS="--steps 3 --warmup 1" bash scripts/r4_ab.sh r04_ab_band_c4.log "default noband" 1 --config C4
