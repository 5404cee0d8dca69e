#!/bin/bash
# launch-shape sweep on C4 (2 kb pairs) and C3 (33 kb pairs): which workgroup shape the host should pick per workload
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; w=d["config"]["workspace"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "wg", w["workgroups"], "thr", w["threads_per_workgroup"], "blk", w["block_levels"], "wsGB", round(w["workspace_bytes"]/1e9,1), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
run C4 "default" SR_X=1
run C4 "128 threads x8/CU" SR_ALIGN_THREADS=128
run C4 "64 threads x16/CU" SR_ALIGN_THREADS=64
run C4 "512 threads x2/CU" SR_ALIGN_THREADS=512
run C4 "hist 16 jobs" SR_HIST_JOBS=16
run C4 "hist 2 jobs" SR_HIST_JOBS=2
run C4 "no reorder" SR_NO_REORDER=1
run C3 "default" SR_X=1
run C3 "512 threads" SR_ALIGN_THREADS=512
run C3 "256 threads" SR_ALIGN_THREADS=256
