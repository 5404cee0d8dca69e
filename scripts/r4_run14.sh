#!/bin/bash
# round 4, run 14: q-gram bound in the alignment kernel's own orientation passes -- parity, then C3 / C5-subset A/B against SR_NO_KBITS=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== parity subset"; timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or several_pairs or randomised or scaled_baseline or parity_cases or orientation or 16bit_ring or c5_three or kernels_and_workgroup or c3" > gpurun_out/r04_t14.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t14.log
grep -q "Memory access fault" gpurun_out/r04_t14.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t14.log | head -5; exit 1; }
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2))"; }
for i in 1 2; do
timeout -k 10 300 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C3 || exit 1
SR_NO_KBITS=1 timeout -k 10 300 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C3-nokbits || exit 1
done
for i in 1 2; do
timeout -k 10 300 python bench.py --config C5 --nseq 64 --steps 2 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C5x64 || exit 1
SR_NO_KBITS=1 timeout -k 10 300 python bench.py --config C5 --nseq 64 --steps 2 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C5x64-nokbits || exit 1
done
