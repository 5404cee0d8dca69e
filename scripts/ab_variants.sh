#!/bin/bash
# A/B/C on one box: built default against the variant libs named as arguments (kernel_build tag printed per run)
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "build", d["config"]["workspace"].get("kernel_build"), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-host-stages > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
for v in "$@"; do
  SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick_$v.log 2>&1; echo "quick $v rc=$?"; tail -1 gpurun_out/quick_$v.log
  grep -q "ALL OK" gpurun_out/quick_$v.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick_$v.log | head; exit 1; }
done
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick default rc=$?"; tail -1 gpurun_out/quick.log
for rep in 1 2; do
  run C2 "default" SR_X=$rep
  for v in "$@"; do run C2 "$v" SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so; done
done
run C4 "default" SR_X=1
for v in "$@"; do run C4 "$v" SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so; done
run C3 "default" SR_X=1
for v in "$@"; do run C3 "$v" SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_$v.so; done
