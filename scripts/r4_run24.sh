#!/bin/bash
# round 4, run 24: a longer fuzz campaign on the final default build (new seed ranges), then the same under the bounds build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
timeout -k 10 900 python scripts/gpu_fuzz.py 260 700 160 60 > gpurun_out/r04_fuzz_long.log 2>&1; echo "fuzz rc=$?"; grep -n "failures\|FAIL\|fault" gpurun_out/r04_fuzz_long.log | head
