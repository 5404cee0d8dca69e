#!/bin/bash
# 32-bit tile (sequences > 32 kb, C5): batched bit-index windows against the variant lib $1; parity first
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
V=$PWD/seqrush_amd/libseqrush_amd_${1:-old32}.so
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], "| ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "build", d["config"]["workspace"].get("kernel_build"), flush=True)
PY
}
run() { cfg=$1; name=$2; shift; shift; env "$@" timeout -k 10 400 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-host-stages $EXTRA > gpurun_out/v.json 2> gpurun_out/v.err && show "$cfg $name" gpurun_out/v.json || { echo "$cfg $name FAILED"; tail -3 gpurun_out/v.err; }; }
timeout -k 10 300 python scripts/gpu_parity_quick.py > gpurun_out/quick.log 2>&1; echo "quick rc=$?"; tail -1 gpurun_out/quick.log
grep -q "ALL OK" gpurun_out/quick.log || { grep -n "MISMATCH\|Error\|error" gpurun_out/quick.log | head; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c5 or int32 or 32 or offsets or ring_u16 or long" > gpurun_out/tp.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/tp.log
EXTRA="--nseq 48"
run C5 "new 48 seqs" SR_X=1
run C5 "old 48 seqs" SEQRUSH_AMD_LIB=$V
run C5 "new 48 seqs" SR_X=2
run C5 "old 48 seqs" SEQRUSH_AMD_LIB=$V
