#!/bin/bash
# round 4, run 13: the Hamming upper bound in the orientation kernel -- parity subset, then C2 / C4 / C3 lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== parity subset"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or several_pairs or randomised or scaled_baseline or parity_cases or orientation or c4 or tree or sparsif" > gpurun_out/r04_t13.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t13.log
grep -q "Memory access fault" gpurun_out/r04_t13.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t13.log | head -5; exit 1; }
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('C2 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'orient', round(r['orient_kernel_ms'], 3))" || exit 1
SR_NO_KBITS=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('C2 nokbits ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'orient', round(r['orient_kernel_ms'], 3))" || exit 1
done
timeout -k 10 300 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('C4 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'orient', round(r['orient_kernel_ms'], 3))" || exit 1
