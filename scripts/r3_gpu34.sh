#!/bin/bash
# C3 tick profile at 1024 / 256 threads per workgroup; counters of single passes of C4 / C2 at several pairs per workgroup
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), {x:k[x] for x in k if k[x]}, flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
run C3 "ticks 1024" SR_PROFILE_TICKS=1 SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_profw.so
run C3 "ticks 256" SR_PROFILE_TICKS=1 SR_ALIGN_THREADS=256
run C3 "plain" SR_X=1
run C2 "nwg 256" SR_NWG=256
run C2 "nwg 64" SR_NWG=64
run C4 "plain" SR_X=1
run C4 "ticks" SR_PROFILE_TICKS=1
