#!/bin/bash
# A/B inside one gpurun call: scripts/r4_ab.sh OUT "name1 name2 ..." [reps] [extra bench args]
# name "default" = seqrush_amd/libseqrush_amd.so, others = libseqrush_amd_<name>.so; interleaved repetitions on one box
out=$1; names=$2; reps=${3:-2}; shift 3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in $(seq 1 $reps); do
  for n in $names; do
    lib=seqrush_amd/libseqrush_amd.so; [ "$n" != default ] && lib=seqrush_amd/libseqrush_amd_$n.so
    SEQRUSH_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2h --no-host-stages ${AB_STEPS:---steps 10 --warmup 3} "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$n FAILED"; tail -3 gpurun_out/ab_tmp.err; continue; }
    python - "$n" "$r" >> gpurun_out/$out <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
r = d["roofline"]; w = d["config"]["workspace"]
print(f"{sys.argv[1]:>10} rep {sys.argv[2]} build={w.get('kernel_build')} ms/step {d['ms_per_step']:.2f} align {r['kernel_ms']:.2f} orient {r.get('orient_kernel_ms')} unite {r['unite']['kernel_ms']:.2f} rowsGB {r['bytes']['rows_counted']/1e9:.1f} rows/ideal {r['bytes']['rows_over_tile_ideal']:.3f} frac {r['frac']:.3f}")
PY
  done
done
cat gpurun_out/$out
