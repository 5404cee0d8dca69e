#!/bin/bash
# round 3, GPU call 4: where a tile's time goes (stamps variant), tick profiles B = 10 / 20, 25-level variant, SQ counters
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), "rowsGB", round(r["bytes"]["rows_counted"]/1e9,1),
      {x:k[x] for x in k if x.startswith(("t","st_","bp_"))})
PY
}
for b in 10 20; do
  SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_stamps.so SR_BLK_LEVELS=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/st_$b.json 2> gpurun_out/st_$b.err || { tail -5 gpurun_out/st_$b.err; exit 1; }
  show "stamps B=$b" gpurun_out/st_$b.json
  SR_PROFILE_TICKS=1 SR_BLK_LEVELS=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/tk_$b.json 2> gpurun_out/tk_$b.err || { tail -5 gpurun_out/tk_$b.err; exit 1; }
  show "ticks B=$b" gpurun_out/tk_$b.json
done
SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_b25.so SR_BLK_LEVELS=25 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/b25.json 2> gpurun_out/b25.err || { tail -5 gpurun_out/b25.err; exit 1; }
show "B=25" gpurun_out/b25.json
SEQRUSH_AMD_LIB=$R/seqrush_amd/libseqrush_amd_b25.so SR_BLK_LEVELS=25 timeout -k 10 200 python scripts/gpu_parity_quick.py > gpurun_out/quick25.log 2>&1; tail -2 gpurun_out/quick25.log
cd /tmp && export TMPDIR=/tmp
for b in 10 20; do
  export SR_BLK_LEVELS=$b
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcA_$b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-stages > $R/gpurun_out/pmcA_$b.out 2> $R/gpurun_out/pmcA_$b.log || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcB_$b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-stages > $R/gpurun_out/pmcB_$b.out 2> $R/gpurun_out/pmcB_$b.log || exit 1
done
cd $R
python - <<'PY'
import csv,glob,collections
for tag in ("pmcA_10","pmcA_20","pmcB_10","pmcB_20"):
    per=collections.defaultdict(float); n=collections.defaultdict(set)
    for path in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "sr_align_blk_kernel" in row["Kernel_Name"]:
                per[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
    print(tag, {k: v/max(1,len(n[k])) for k,v in per.items()})
PY
