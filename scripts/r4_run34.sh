#!/bin/bash
# round 4, run 34: sensitivity builds -- +20 VALU instructions per tile level (+~10 % of a tile's VALU), +4 / +8 cold row loads per
# tile (+~13 / 26 % of a tile's row loads) -- against the same source without them; one box, alternating, C2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; w = d['config']['workspace']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'rowsGB', round(r['bytes']['rows_counted']/1e9,1), 'loadedGB', round(r['bytes']['rows_loaded']/1e9,1), 'build', w['kernel_build'])"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2; do
for v in inj0 injv injl injl8; do
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so timeout -k 10 300 $B 2>/dev/null | line C2 || exit 1
done
done
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_injv.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_c2_parity" 2>&1 | tail -n 1
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_injl8.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_c2_parity" 2>&1 | tail -n 1
