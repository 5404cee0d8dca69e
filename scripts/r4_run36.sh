#!/bin/bash
# round 4, run 36: lean pass (cell tallies in LDS, the pass's level opaque to loop strength reduction: 132 -> 99 spilled VGPRs,
# 247 -> 196 scratch instructions) against the same source with -DSR_LEAN_PASS=0; single-instance experiment builds, one box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; w = d['config']['workspace']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'cells', d['kernels']['wf_cells'], 'build', w['kernel_build'])"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2 3; do
for v in p0 p1; do
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so timeout -k 10 300 $B 2>/dev/null | line C2 || exit 1
done
done
B4="python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages"
for v in p0 p1 p0 p1; do
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so timeout -k 10 300 $B4 2>/dev/null | line C4 || exit 1
done
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_p1.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_c2_parity" 2>&1 | tail -n 1
