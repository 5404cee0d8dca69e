#!/bin/bash
# what the driver runs at round end, in its order: smoke(), the default bench line, the -m gpu suite
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
SECONDS=0; timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; echo "wall ${SECONDS}s"; tail -1 gpurun_out/bench_default.json | cut -c1-400
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "bench --gpus 2 on a 1-GPU box rc=$? (expected to fail cleanly: only one device)"; tail -2 gpurun_out/bench_g2.err | cut -c1-300
