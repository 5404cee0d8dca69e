#!/bin/bash
# round 4, run 29: tile statistics (experiment build) on C2 and C4
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
export SEQRUSH_AMD_LIB=$PWD/seqrush_amd/libseqrush_amd_stats.so
timeout -k 10 300 python scripts/tile_stats.py C2 2>/dev/null | tail -1 | tee gpurun_out/r04_tile_stats.log
timeout -k 10 300 python scripts/tile_stats.py C4 2>/dev/null | tail -1 | tee -a gpurun_out/r04_tile_stats.log
