#!/bin/bash
# the whole -m gpu suite + smoke + default bench
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/tfull.log 2>&1
echo "pytest rc=$?"; tail -14 gpurun_out/tfull.log
