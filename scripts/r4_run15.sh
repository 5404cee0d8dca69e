#!/bin/bash
# round 4, run 15: unite fused into the blocked alignment kernel (sr_ctx_run) -- parity, then C2 A/B against SR_NO_FUSED_UNITE=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== parity subset"; timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "smoke or c1_8x1kb or c2_subset or c2_full or several_pairs or randomised or scaled_baseline or parity_cases or orientation or 16bit_ring or c5_three or kernels_and_workgroup or c3 or batch or arena or paf or label or shard" > gpurun_out/r04_t15.log 2>&1; rc=$?; tail -3 gpurun_out/r04_t15.log
grep -q "Memory access fault" gpurun_out/r04_t15.log && { echo FAULT; exit 1; }
[ $rc -ne 0 ] && { grep -E "Error|assert" gpurun_out/r04_t15.log | head -5; exit 1; }
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), 'orient', r.get('orient_kernel_ms'), 'unite', r['unite'].get('kernel_ms'), 'united', d['kernels']['united_bases'], 'fused', d['config']['workspace'].get('fused_unite'))"; }
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C2 || exit 1
SR_NO_FUSED_UNITE=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C2-unfused || exit 1
done
timeout -k 10 300 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C3 || exit 1
SR_NO_FUSED_UNITE=1 timeout -k 10 300 python bench.py --config C3 --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C3-unfused || exit 1
timeout -k 10 300 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C4 || exit 1
SR_NO_FUSED_UNITE=1 timeout -k 10 300 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | line C4-unfused || exit 1
