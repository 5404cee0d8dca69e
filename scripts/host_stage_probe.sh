#!/bin/bash
# host stages around the device path for C2 and C4
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for cfg in C2 C4; do
  timeout -k 10 300 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/v.json 2> gpurun_out/v.err || { tail -3 gpurun_out/v.err; exit 1; }
  python - $cfg <<'PY'
import json,sys
d=json.loads(open("gpurun_out/v.json").read().strip().split("\n")[-1])
print(sys.argv[1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d["host_stages_ms"].items()}, flush=True)
PY
done
