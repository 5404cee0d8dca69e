#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== parity subset (default)"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or several_pairs or requeue or randomised or scaled_baseline or parity_cases or 16bit_ring or c5_three or orientation or full_size_c3 or deep_levels or full_size_c2_parity" > gpurun_out/r04_t9.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t9.log
grep -q "Memory access fault" gpurun_out/r04_t9.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t9.log | head -5; exit 1; }
rm -f gpurun_out/r04_ab_mask.log; bash scripts/r4_ab.sh r04_ab_mask.log "default nomask" 2
rm -f gpurun_out/r04_ab_ori_c3.log; bash scripts/r4_ab.sh r04_ab_ori_c3.log "default ori8" 2 --config C3
rm -f gpurun_out/r04_ab_ori_c5.log; AB_STEPS="--steps 3 --warmup 1" bash scripts/r4_ab.sh r04_ab_ori_c5.log "default ori8" 1 --config C5 --nseq 32
