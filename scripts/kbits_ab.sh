#!/bin/bash
# q-gram bound of the reverse orientation's score (a.kbits) against SR_NO_KBITS=1, same library: parity first
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "orient", round(r.get("orient_kernel_ms") or 0,3), "orient cells", r.get("orient_level_diagonals"), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "orient or c1 or c2_subset or rc or reverse or several_pairs or scaled" > gpurun_out/tp.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tp.log
for rep in 1 2; do run C2 "kbits" SR_X=$rep; run C2 "no kbits" SR_NO_KBITS=1; done
run C4 "kbits" SR_X=1; run C4 "no kbits" SR_NO_KBITS=1
