#!/bin/bash
# usage: bench_sweep.sh "ENV1=.. ENV2=.." "ENV.." ...   -> one short bench line per configuration
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernels']
    print('ms/step %.1f  kernel %.1f  pass %.2fG ph2 %.2fG ori %.2fG base %.2fG pair %.2fG' % (d['ms_per_step'], d['roofline']['kernel_ms'], k['tk_pass']/1e9, k['tk_phase2']/1e9, k['ticks_orientation']/1e9, k['ticks_base']/1e9, k['ticks_pair']/1e9))
except Exception as e: print('FAILED', e)
"
done
