#!/bin/bash
# occupancy / block-depth variants, same box, same call
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]; w=d["config"]["workspace"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "wg", w["workgroups"], "blk", w["block_levels"], "lds", w["lds_dynamic_bytes"])
PY
}
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$name" gpurun_out/v.json || { echo "$name FAILED"; tail -3 gpurun_out/v.err; }; }
L=$R/seqrush_amd
run "default(kb20,b10)" SR_X=1
run "k10 b10" SEQRUSH_AMD_LIB=$L/libseqrush_amd_k10.so SR_STATIC_LDS_KB=20
run "k15 b15" SEQRUSH_AMD_LIB=$L/libseqrush_amd_k15.so SR_BLK_LEVELS=15 SR_STATIC_LDS_KB=24
run "k15 b10" SEQRUSH_AMD_LIB=$L/libseqrush_amd_k15.so SR_STATIC_LDS_KB=24
run "w5k15 b15 5wg" SEQRUSH_AMD_LIB=$L/libseqrush_amd_w5k15.so SR_BLK_LEVELS=15 SR_STATIC_LDS_KB=23 SR_WG_PER_CU=5
run "w5k15 b15 4wg" SEQRUSH_AMD_LIB=$L/libseqrush_amd_w5k15.so SR_BLK_LEVELS=15 SR_STATIC_LDS_KB=23
run "w5k10 b10 5wg" SEQRUSH_AMD_LIB=$L/libseqrush_amd_w5k10.so SR_STATIC_LDS_KB=20 SR_WG_PER_CU=5
run "default again" SR_X=2
SEQRUSH_AMD_LIB=$L/libseqrush_amd_w5k15.so SR_BLK_LEVELS=15 SR_STATIC_LDS_KB=23 SR_WG_PER_CU=5 timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick_w5.log 2>&1; tail -1 gpurun_out/quick_w5.log
