#!/bin/bash
# round 3, GPU call 2: counter list, workgroups-in-flight sweep, tick profile, oracle step timing
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1 || rocprofv3 --list-avail > $R/gpurun_out/counters_list.txt 2>&1 || true)
cd $R
for n in 1024 896 768 640 512 384; do
  SR_NWG=$n timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/nwg_$n.json 2> gpurun_out/nwg_$n.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/nwg_$n.json").read().strip().split("\n")[-1])
print("NWG $n ms/step", round(d["ms_per_step"],2), "align", round(d["roofline"]["kernel_ms"],2))
PY
done
SR_PROFILE_TICKS=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-stages > gpurun_out/ticks.json 2> gpurun_out/ticks.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ticks.json").read().strip().split("\n")[-1])
k=d["kernels"]; print({x:k[x] for x in k if x.startswith("t")}, d["roofline"]["kernel_ms"])
PY
python - <<'PY'
import sys,time,os,subprocess
code='''
import sys,time,os; sys.path.insert(0,"tests"); sys.path.insert(0,".")
import oracle_binding as ob
from seqrush_amd import synth
recs=synth.config_c2(16)
o=ob.OracleSeqRush(records=recs)
pairs=[(q,t) for q in range(16) for t in range(16)][:96]
for th in (1,16):
    op=ob.default_params(); op.threads=th
    n=12 if th==1 else 96
    t=time.time(); o.align_list_collect(op,pairs[:n],unite=False); dt=time.time()-t
    print("oracle", os.environ.get("SRO_ORACLE_SCALAR_STEP"), "threads", th, round(n/dt,1), "pairs/s")
'''
for m in ("0","1"):
    subprocess.run([sys.executable,"-c",code],env=dict(os.environ,SRO_ORACLE_SCALAR_STEP=m))
PY
