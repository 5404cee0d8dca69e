#!/bin/bash
# round 4 final measurements on one box: PMC / stats profile of the bench workload, bench lines of C2..C5, shard probe
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/profile_round.sh r04 > gpurun_out/profile_r04.log 2>&1 || { tail -20 gpurun_out/profile_r04.log; echo "profile failed"; }
tail -3 gpurun_out/profile_r04.log | cut -c1-300
cp gpurun_out/prof_r04/r04_counters.json profiles/r04_counters.json 2>/dev/null
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_C2.json 2> gpurun_out/r04_bench_C2.err; echo "C2 rc=$?"
timeout -k 10 600 python bench.py --config C3 --steps 10 --warmup 2 > gpurun_out/r04_bench_C3.json 2> gpurun_out/r04_bench_C3.err; echo "C3 rc=$?"
timeout -k 10 600 python bench.py --config C4 --steps 5 --warmup 1 > gpurun_out/r04_bench_C4.json 2> gpurun_out/r04_bench_C4.err; echo "C4 rc=$?"
timeout -k 10 900 python bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline --no-h2h > gpurun_out/r04_bench_C5.json 2> gpurun_out/r04_bench_C5.err; echo "C5 rc=$?"
timeout -k 10 300 python scripts/shard_probe.py "SR_ALIGN_THREADS=512" > gpurun_out/r04_shard_probe.log 2>&1; grep shards gpurun_out/r04_shard_probe.log | head -6
python - <<'PY'
import json
for c in ("C2","C3","C4","C5"):
    try:
        d=json.loads(open(f"gpurun_out/r04_bench_{c}.json").read().strip().split("\n")[-1])
        r=d["roofline"]; w=d["config"]["workspace"]
        print(c, "ms/step", round(d["ms_per_step"],2), "pairs/s", round(d["value"],1), "gcups", round(d["gcups"],1), "align", round(r["kernel_ms"],2), "frac", round(r["frac"],3), "model_frac", round(r["model_frac"],3), "bound", r["bound"], "h2h", d.get("h2h_ms"), "wsGB", round(w["workspace_bytes"]/1e9,1), "traffic", r.get("traffic"), d.get("host_stages_ms"))
    except Exception as e:
        print(c, "failed", e)
PY
