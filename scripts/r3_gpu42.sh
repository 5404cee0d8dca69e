#!/bin/bash
# orientation with batched bit-index windows (separate kernel and in-kernel blocks): parity, then C3 / C5-subset / C2 against the variant lib $1
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
V=$PWD/seqrush_amd/libseqrush_amd_${1:-oldori}.so
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "orient", r.get("orient_kernel_ms"), "build", d["config"]["workspace"].get("kernel_build"), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-host-stages $EXTRA > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "orient or c3 or c5_three or scaled or raw_byte or several_pairs or kernels_and_workgroup" > gpurun_out/tp.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tp.log
run C3 "new" SR_X=1
run C3 "old" SEQRUSH_AMD_LIB=$V
run C3 "new" SR_X=2
run C3 "old" SEQRUSH_AMD_LIB=$V
run C2 "new (orientation kernel changed in both libs)" SR_X=1
run C2 "in-kernel new" SR_PREORIENT=0
run C2 "in-kernel old" SR_PREORIENT=0 SEQRUSH_AMD_LIB=$V
