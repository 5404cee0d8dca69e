#!/bin/bash
# Round profile of the bench workload (C2, N=1) on the GPU box: kernel-trace stats + PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ counters; one pass each: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Never combined with
# sys/runtime traces.  Output under gpurun_out/prof_$TAG/, summary JSON = gpurun_out/prof_$TAG/${TAG}_counters.json
# (copy into profiles/).   usage: scripts/profile_round.sh r02 [bench args...]
set -e
TAG=${1:-r04}; shift || true
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-h2h --no-host-stages $*"
python3 bench.py $ARGS > "$OUT/bench_plain.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_under_stats.json" 2> "$OUT/stats.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/fetch.out" 2> "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/write.out" 2> "$OUT/write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq1" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/sq1.out" 2> "$OUT/sq1.log"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/sq2.out" 2> "$OUT/sq2.log"
# memory-side requests of the L2 by destination (round 3): DRAM-destined reads / writes in 32-byte units (exact bytes, no
# calibration factor), all reads, L2 hits / misses.  gfx950 exposes no Infinity-Cache hit counter: "DRAM-destined" includes them.
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/tcc1" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/tcc1.out" 2> "$OUT/tcc1.log" || true
rocprofv3 --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/tcc2" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/tcc2.out" 2> "$OUT/tcc2.log" || true
# round 4: instruction cache and LDS bank conflicts (is the 180 KB kernel fetch-bound? do the window reads conflict?)
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d "$OUT/sq3" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/sq3.out" 2> "$OUT/sq3.log" || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d "$OUT/sq4" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-h2h --no-host-stages $* > "$OUT/sq4.out" 2> "$OUT/sq4.log" || true
cd "$ROOT"
SR_PROFILE_HOST=$(hostname) SR_PROFILE_GIT=$(cat "$ROOT/.git_head" 2>/dev/null || echo unknown) python3 scripts/profile_collect.py "$OUT" "$TAG" > "$OUT/${TAG}_counters.json"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
cat "$OUT/${TAG}_counters.json"
