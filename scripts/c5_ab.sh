#!/bin/bash
# C5 subset (48 x 50 kb = 2304 pairs of the 32-bit tile): built default against the variant lib $1, interleaved
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
run() { name=$1; shift; env "$@" timeout -k 10 400 python bench.py --config C5 --nseq 48 --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "C5/48 $name" gpurun_out/v.json || { echo "$name FAILED"; tail -3 gpurun_out/v.err; }; }
for rep in 1 2 3; do run default SR_X=$rep; run $1 SEQRUSH_AMD_LIB=$V; done
