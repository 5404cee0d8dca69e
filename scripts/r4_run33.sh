#!/bin/bash
# round 4, run 33: pair latency against workgroups per CU (SR_WG_PER_CU = 1..4, all 4 096 pairs of C2), one box; bound label of the line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
for n in 4 3 2 1 4; do
SR_WG_PER_CU=$n timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; w = d['config']['workspace']
n = w['workgroups_per_cu']; ms = r['kernel_ms']
print('wg/cu', n, 'wgs', w['workgroups'], 'align ms', round(ms, 2), 'pair latency ms', round(ms * n * 256 / 4096, 2), 'bound', r['bound'], '|', r['bound_basis']['note'][:160])" || exit 1
done
