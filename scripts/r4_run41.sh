#!/bin/bash
# round 4, run 41: profile (stats + PMC passes) and the 20-step C2 line of the shipped build
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/profile_round.sh r04 > gpurun_out/profile_r04.log 2>&1 || { tail -20 gpurun_out/profile_r04.log; echo "profile failed"; }
tail -2 gpurun_out/profile_r04.log | cut -c1-200
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_C2.json 2> gpurun_out/r04_bench_C2.err; echo "C2 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_C2.json").read().strip().split("\n")[-1]); r=d["roofline"]
print("C2 ms/step", round(d["ms_per_step"],2), "pairs/s", round(d["value"],1), "align", round(r["kernel_ms"],2), "frac", round(r["frac"],3), "bound", r["bound"], "h2h", d.get("h2h_ms"), "traffic", r.get("traffic"), "cpu", d["cpu_baseline"]["value"], d.get("telemetry"))
PY
