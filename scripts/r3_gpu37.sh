#!/bin/bash
# A/B: extension with bit-index windows / v_ffbl / ext == window flags (variant lib $1) against the built default
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
V=$PWD/seqrush_amd/libseqrush_amd_${1:-ext2}.so
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "build", d["config"]["workspace"].get("kernel_build"), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
SEQRUSH_AMD_LIB=$V timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
run C2 "default" SR_X=1
run C2 "variant" SEQRUSH_AMD_LIB=$V
run C2 "default" SR_X=2
run C2 "variant" SEQRUSH_AMD_LIB=$V
run C4 "default" SR_X=1
run C4 "variant" SEQRUSH_AMD_LIB=$V
run C3 "default" SR_X=1
run C3 "variant" SEQRUSH_AMD_LIB=$V
