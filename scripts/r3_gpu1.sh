#!/bin/bash
# round 3, GPU call 1: new full-size parity tests + multi-rank launch paths + bench line + counter list
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(rocprofv3 --list-counters > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1 || true)
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
  -k "full_size_c2_parity or full_size_c4_parity or multi_rank_bench or multi_gpu_cli or failing_rank or smoke" \
  --durations=10 > gpurun_out/t1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t1.log
tail -25 gpurun_out/t1.log
timeout -k 10 300 python bench.py > gpurun_out/bench1.json 2> gpurun_out/bench1.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/bench1.json
