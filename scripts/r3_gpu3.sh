#!/bin/bash
# round 3, GPU call 3: rolled 20-level tile -- quick parity, bench B = 10 vs 20, then the parity test file
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -12 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || exit 1
for b in 10 20; do
  SR_BLK_LEVELS=$b timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/blk_$b.json 2> gpurun_out/blk_$b.err || { tail -5 gpurun_out/blk_$b.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/blk_$b.json").read().strip().split("\n")[-1])
r=d["roofline"]
print("B $b ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "rows GB", round(r["bytes"]["rows_counted"]/1e9,1), "ldiag", r["wf_level_diagonals"], "blk", d["config"]["workspace"]["block_levels"], "passes", d["kernels"]["bp_passes"])
PY
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size_c4_parity and not full_size_c5" > gpurun_out/t3.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/t3.log
