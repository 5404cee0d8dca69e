#!/bin/bash
# round 4, run 39: control section's per-lane addresses formed per pass from an opaque thread index (SR_LEAN_CTL=1: 97 -> 94 spilled VGPRs,
# 197 -> 188 scratch instructions) against SR_LEAN_CTL=0; single-instance experiment builds, one box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; w = d['config']['workspace']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'build', w['kernel_build'])"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2 3; do
for v in q0 q1; do
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so timeout -k 10 300 $B 2>/dev/null | line C2 || exit 1
done
done
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_q1.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_c2_parity" 2>&1 | tail -n 1
