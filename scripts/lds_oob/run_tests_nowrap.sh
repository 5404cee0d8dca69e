#!/bin/bash
# the parity tests that align several pairs per workgroup, the penalty-set cases and the randomised sets on the build
# WITHOUT LDS address wrapping and with the packed tile on the generic instance (oobA)
cd $GRAFT_REPO_ROOT
export SEQRUSH_AMD_LIB=$GRAFT_REPO_ROOT/seqrush_amd/libseqrush_amd_oobA.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "several_pairs or parity_cases or randomised or all_align_kernels or c2_subset or scaled_baseline" 2>&1 | tail -4
for i in 1 2 3; do SR_NWG=$i SR_BLK_LEVELS=5 SR_ALIGN_THREADS=256 timeout -k 5 120 python scripts/pair_probe.py 6 all 2>&1 | tail -1; done
