#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
echo "== parity (default)"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or several_pairs or requeue or randomised or scaled_baseline or parity_cases or 50kb or 16bit_ring or c5_three or offset_width" 2>&1 | tail -4
echo "== parity under the bounds build (16-bit ring, 50 kb)"; SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_bounds.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "16bit_ring or c5_three or several_pairs" 2>&1 | tail -4
rm -f gpurun_out/r04_ab_b3.log; bash scripts/r4_ab.sh r04_ab_b3.log "default b3 tight" 2
rm -f gpurun_out/r04_ab_c5.log; bash scripts/r4_ab.sh r04_ab_c5.log "default nopku16" 1 --config C5 --nseq 24
echo "== ticks"; for nwg in "" 256; do SR_NWG=$nwg SR_PROFILE_TICKS=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages --no-h2h > gpurun_out/v.json 2> gpurun_out/v.err && python - "$nwg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/v.json").read().strip().split("\n")[-1])
r=d["roofline"]; k=d["kernels"]; tp=k.get("ticks_pair") or 1
print("NWG", sys.argv[1] or "default", "ms/step", round(d["ms_per_step"],2), "align", round(r["kernel_ms"],2), {x: round(100.0*k[x]/tp,1) for x in k if x.startswith(("tk_","ticks_b","ticks_br")) and k[x]}, "ticks_pair_ms_per_pair", tp/100e3/4096)
PY
done
