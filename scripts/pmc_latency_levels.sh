#!/bin/bash
# PMC: instruction counts and in-flight levels (average LDS / VMEM latency) of the alignment kernel, default vs variant lib $1
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
V=$R/seqrush_amd/libseqrush_amd_${1:-ext2}.so
cd /tmp && export TMPDIR=/tmp
pass() { # name, counters..., (env SEQRUSH_AMD_LIB optional through $LIBV)
  name=$1; shift
  out=$R/gpurun_out/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-stages > $out.out 2> $out.log || { echo "$name failed"; tail -5 $out.log; return 1; }
  python3 - $out "$name" <<'PY'
import sys,glob,csv,collections
d=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sr_align_blk_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print(sys.argv[2], {k: v for k,v in sorted(d.items())}, flush=True)
PY
}
pass A_def SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass B_def SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
export SEQRUSH_AMD_LIB=$V
pass A_var SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
