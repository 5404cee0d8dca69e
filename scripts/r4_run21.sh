#!/bin/bash
# round 4, run 21: three-wave workgroups, five per CU (-DSR_NT192 experiment build) against four waves x four
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
export SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_nt192.so
echo "== parity (192 x 5)"; SR_ALIGN_THREADS=192 SR_WG_PER_CU=5 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or randomised" > gpurun_out/r04_t21.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t21.log
grep -q "Memory access fault" gpurun_out/r04_t21.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t21.log | head -5; exit 1; }
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; w = d['config']['workspace']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'thr', w['threads_per_workgroup'], 'wg/cu', w['workgroups_per_cu'], 'wgs', w['workgroups'], 'build', w['kernel_build'])"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2; do
timeout -k 10 300 $B 2>/dev/null | line C2-256x4 || exit 1
SR_ALIGN_THREADS=192 SR_WG_PER_CU=5 timeout -k 10 300 $B 2>/dev/null | line C2-192x5 || exit 1
SR_ALIGN_THREADS=192 SR_WG_PER_CU=4 timeout -k 10 300 $B 2>/dev/null | line C2-192x4 || exit 1
done
B4="python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2; do
timeout -k 10 300 $B4 2>/dev/null | line C4-256x4 || exit 1
SR_ALIGN_THREADS=192 SR_WG_PER_CU=5 timeout -k 10 300 $B4 2>/dev/null | line C4-192x5 || exit 1
done
