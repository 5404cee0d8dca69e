#!/bin/bash
# quick parity + bench (plain and tick profile) for each "ENV=.. ENV=.." variant given as an argument; then optional pytest
# usage: scripts/r3_probe.sh [--tests] "SR_X=1" "SR_BLK_LEVELS=20" ...
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
TESTS=0; if [ "$1" == "--tests" ]; then TESTS=1; shift; fi
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "wsGB", round(d["config"]["workspace"]["workspace_bytes"]/1e9,1),
      {x:k[x] for x in k if x.startswith(("tk","ticks_b","ticks_p","bp_e","bp_c")) and k[x]})
PY
}
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/f.json 2> gpurun_out/f.err || { tail -5 gpurun_out/f.err; exit 1; }
  show "plain $v" gpurun_out/f.json
  env $v SR_PROFILE_TICKS=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/ft.json 2> gpurun_out/ft.err || { tail -5 gpurun_out/ft.err; exit 1; }
  show "ticks $v" gpurun_out/ft.json
done
if [ $TESTS == 1 ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size_c4_parity and not full_size_c5" > gpurun_out/tp.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/tp.log
fi
