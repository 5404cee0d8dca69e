cd $GRAFT_REPO_ROOT
for cfg in "C3" "C2 --nseq 23"; do
SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_profwide.so SR_PROFILE_TICKS=1 timeout -k 10 200 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages --no-h2h > gpurun_out/v.json 2> gpurun_out/v.err && python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/v.json").read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]; tp=k.get("ticks_pair") or 1; w=d["config"]["workspace"]
print(sys.argv[1], "threads", w["threads_per_workgroup"], "wgs", w["workgroups"], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "passes/pair", k["bp_passes"]/d["config"]["pairs_total"], {x: round(100.0*k[x]/tp,1) for x in k if x.startswith(("tk_","ticks_")) and k[x]}, "ms per pair (ticks)", tp/100e3/d["config"]["pairs_total"])
PY
done
