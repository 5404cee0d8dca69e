#!/bin/bash
# counters moved to LDS + 10-bit walk-order field: parity, the deep-scope tests, counters of C2 / C3 / C4, generic instance time
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), {x:k[x] for x in ("wf_steps","base_segments","breakpoint_searches","bp_passes","bp_rounds")}, flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep_scope or deep_ring or without_blocked or several_pairs" > gpurun_out/tp.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/tp.log
run C2 "plain" SR_X=1
run C2 "plain again" SR_X=2
run C2 "generic B=5" SR_BLK_LEVELS=5
run C3 "plain" SR_X=1
run C4 "plain" SR_X=1
