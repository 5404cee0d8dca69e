#!/bin/bash
# round 4, run 30: merged tail tiles (the last tiles of a segment's two aligners in one wave) -- parity, statistics, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M=$PWD/seqrush_amd/libseqrush_amd_mrg.so; N=$PWD/seqrush_amd/libseqrush_amd_nomrg.so
echo "== parity (default build: merged tails on)"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or randomised or full_size_c2 or several_pairs or parity_cases or 16bit_ring or c5_three or 32bit_searches or requeue or deep_levels or c3" > gpurun_out/r04_t30.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t30.log
grep -q "Memory access fault" gpurun_out/r04_t30.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t30.log | head -5; exit 1; }
SEQRUSH_AMD_LIB=$M timeout -k 10 300 python scripts/tile_stats.py C2 2>/dev/null | tail -n 1 | cut -c1-700
SEQRUSH_AMD_LIB=$N timeout -k 10 300 python scripts/tile_stats.py C2 2>/dev/null | tail -n 1 | cut -c1-700
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; w = d['config']['workspace']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'rowsGB', round(r['bytes']['rows_counted']/1e9,1), 'build', w['kernel_build'])"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2 3; do
SEQRUSH_AMD_LIB=$N timeout -k 10 300 $B 2>/dev/null | line C2 || exit 1
SEQRUSH_AMD_LIB=$M timeout -k 10 300 $B 2>/dev/null | line C2 || exit 1
done
B4="python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2; do
SEQRUSH_AMD_LIB=$N timeout -k 10 300 $B4 2>/dev/null | line C4 || exit 1
SEQRUSH_AMD_LIB=$M timeout -k 10 300 $B4 2>/dev/null | line C4 || exit 1
done
