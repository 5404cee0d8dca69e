#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), {x:k[x] for x in k if x.startswith(("tk","ticks_b","ticks_p")) and k[x]})
PY
}
for v in "SR_X=1" "SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_maklds.so" "SR_X=2" "SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_maklds.so"; do
  env $v timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/f.json 2> gpurun_out/f.err || { tail -5 gpurun_out/f.err; exit 1; }
  show "plain $v" gpurun_out/f.json
done
for v in "SR_X=1" "SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_maklds.so"; do
  env $v SR_PROFILE_TICKS=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/ft.json 2> gpurun_out/ft.err || { tail -5 gpurun_out/ft.err; exit 1; }
  show "ticks $v" gpurun_out/ft.json
done
SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_maklds.so timeout -k 10 300 python scripts/gpu_parity_quick.py 2>&1 | tail -1
