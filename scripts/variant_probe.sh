#!/bin/bash
# one-GPU A/B of launch-shape knobs on the C2 bench (kernel ms from hipEvents); usage: scripts/variant_probe.sh out.log
out=${1:-gpurun_out/variants.log}
: > $out
run() { echo "== $*" >> $out; env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>>$out.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.2f kernel %.2f orient %.2f unite %.2f' % (d['ms_per_step'], r['kernel_ms'], r.get('orient_kernel_ms') or 0, r.get('unite_kernel_ms') or 0))" >> $out; }
run SR_X=0
run SR_ALIGN_THREADS=512
run SR_WG_PER_CU=3
run SR_WG_PER_CU=2
run SR_WG_PER_CU=2 SR_ALIGN_THREADS=512
run SR_BLK_LEVELS=5
run SR_PREORIENT=0
cat $out
