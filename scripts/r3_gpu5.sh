#!/bin/bash
# round 3, GPU call 5: band-based breakpoint tests, wave emission, packed base-case history
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "rowsGB", round(r["bytes"]["rows_counted"]/1e9,1), "wsGB", round(d["config"]["workspace"]["workspace_bytes"]/1e9,1),
      {x:k[x] for x in k if x.startswith(("t","bp_")) and k[x]})
PY
}
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -3 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
for b in 10 20; do
  SR_BLK_LEVELS=$b timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/e_$b.json 2> gpurun_out/e_$b.err || { tail -5 gpurun_out/e_$b.err; exit 1; }
  show "plain B=$b" gpurun_out/e_$b.json
  SR_PROFILE_TICKS=1 SR_BLK_LEVELS=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/etk_$b.json 2> gpurun_out/etk_$b.err || { tail -5 gpurun_out/etk_$b.err; exit 1; }
  show "ticks B=$b" gpurun_out/etk_$b.json
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size_c4_parity and not full_size_c5" > gpurun_out/t5.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/t5.log
