#!/bin/bash
# round 4, run 18: knob sweep on the final build (one box, alternating with the default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$1 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), r['kernel_ms_min_max'], 'rowsGB', round(r['bytes']['rows_counted']/1e9,1), 'knobs', d['config']['workspace'].get('knobs'))"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --no-host-stages"
for i in 1 2; do
timeout -k 10 300 $B 2>/dev/null | line default || exit 1
SR_LAZY_ID=0 timeout -k 10 300 $B 2>/dev/null | line eagerID || exit 1
SR_WG_PER_CU=3 timeout -k 10 300 $B 2>/dev/null | line wg3 || exit 1
SR_NO_REORDER=1 timeout -k 10 300 $B 2>/dev/null | line noreorder || exit 1
SR_HIST_JOBS=8 timeout -k 10 300 $B 2>/dev/null | line hist8 || exit 1
done
