#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; w=d["config"]["workspace"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "wg", w["workgroups"], "thr", w["threads_per_workgroup"], "wsGB", round(w["workspace_bytes"]/1e9,1))
PY
}
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$name" gpurun_out/v.json || { echo "$name FAILED"; tail -3 gpurun_out/v.err; }; }
run "default" SR_X=1
run "128 threads x8/CU" SR_ALIGN_THREADS=128
run "64 threads x16/CU" SR_ALIGN_THREADS=64
run "512 threads x2/CU" SR_ALIGN_THREADS=512
run "hist 16 jobs" SR_HIST_JOBS=16
run "no reorder" SR_NO_REORDER=1
run "preorient 0" SR_PREORIENT=0
run "default" SR_X=2
