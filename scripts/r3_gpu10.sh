#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python scripts/shard_probe.py "SR_ALIGN_THREADS=256;SR_ALIGN_THREADS=256,SR_BLK_LEVELS=20;SR_ALIGN_THREADS=256,SR_BLK_LEVELS=15;SR_ALIGN_THREADS=512" 2>&1 | tee gpurun_out/shard_r3a.log | grep shards
