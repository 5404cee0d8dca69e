#!/bin/bash
# tick profile incl. the control section's parts (variant lib $1)
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
V=$PWD/seqrush_amd/libseqrush_amd_${1:-ctlprof}.so
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
tp=k.get("ticks_pair") or 1
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), d["config"]["workspace"].get("kernel_build"), {x: round(100.0*k[x]/tp,1) for x in k if x.startswith(("tk_","ticks_b","ticks_br")) and k[x]}, flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
run C2 "ticks" SR_PROFILE_TICKS=1 SEQRUSH_AMD_LIB=$V
run C2 "ticks 512 pairs/4 waves" SR_PROFILE_TICKS=1 SEQRUSH_AMD_LIB=$V SR_NWG=256
run C4 "ticks" SR_PROFILE_TICKS=1 SEQRUSH_AMD_LIB=$V
