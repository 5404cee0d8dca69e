#!/bin/bash
# quick parity of a set of library builds on one box: scripts/r4_quick.sh "name1 name2 ..."  (default = the default library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep "\[build\]"
for n in $1; do
  lib=seqrush_amd/libseqrush_amd.so; [ "$n" != default ] && lib=seqrush_amd/libseqrush_amd_$n.so
  echo "== $n"
  SEQRUSH_AMD_LIB=$PWD/$lib timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_8x1kb or c2_subset or several_pairs or requeue or randomised or scaled_baseline or parity_cases" 2>&1 | tail -4
done
