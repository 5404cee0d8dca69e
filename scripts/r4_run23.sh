#!/bin/bash
# round 4, run 23: bench line with clock / power telemetry, three runs on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('C2 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), r['kernel_ms_min_max'], d.get('telemetry'))" || exit 1
done
timeout -k 10 300 python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('C4 ms/step', round(d['ms_per_step'], 2), 'align', round(r['kernel_ms'], 2), d.get('telemetry'))"
