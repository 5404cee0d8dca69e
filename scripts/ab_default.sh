#!/bin/bash
# repeated interleaved A/B of one variant lib ($1) against the built default on C2 (4x) and C4 (2x)
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
V=$PWD/seqrush_amd/libseqrush_amd_$1.so
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "build", d["config"]["workspace"].get("kernel_build"), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
SEQRUSH_AMD_LIB=$V timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick $1 rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
run C2 "warm" SR_X=0
for rep in 1 2 3 4; do run C2 "default" SR_X=$rep; run C2 "$1" SEQRUSH_AMD_LIB=$V; done
for rep in 1 2; do run C4 "default" SR_X=$rep; run C4 "$1" SEQRUSH_AMD_LIB=$V; done
