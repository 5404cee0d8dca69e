#!/bin/bash
# round 4, run 25: fuzz campaign, ten times run 24 (new seeds), default build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
for m in 120 180 240 300 360; do
timeout -k 10 400 python scripts/gpu_fuzz.py $((700 + (m - 120) * 14)) $((700 + (m - 60) * 14)) 0 $m > gpurun_out/r04_fuzz_long_$m.log 2>&1; echo "fuzz m=$m rc=$?"; grep -n "failures\|FAIL\|fault" gpurun_out/r04_fuzz_long_$m.log | head -5
done
