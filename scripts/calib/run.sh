#!/bin/bash
# Calibration pass on the GPU box (see calib.hip): plain run for times / VALU rates, then one rocprofv3 pass per
# TCC counter (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output: gpurun_out/calib/r02_calibration.json
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/calib
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/calib" scripts/calib/calib.hip
"$OUT/calib" "$OUT/plain.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- "$OUT/calib" "$OUT/under_fetch.json" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- "$OUT/calib" "$OUT/under_write.json" > "$OUT/write.log" 2>&1
cd "$ROOT"
python3 scripts/calib/collect.py "$OUT" > "$OUT/r02_calibration.json"
cat "$OUT/r02_calibration.json"
