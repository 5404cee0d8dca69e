#!/bin/bash
# the round-2 symptom ("second pair of a workgroup fails") on builds that do not wrap the LDS window addresses:
#   oobA: no wrap (addresses only word-aligned), packed tile on the generic 5-level instance
#   oobB: oobA + cells below 0 gain nothing from their windows
#   oobC: wrap as shipped, packed tile on the generic instance
cd $GRAFT_REPO_ROOT
for v in oobC oobA oobB; do
  echo "== $v"
  SEQRUSH_AMD_LIB=$GRAFT_REPO_ROOT/seqrush_amd/libseqrush_amd_$v.so SR_NWG=1 SR_BLK_LEVELS=5 SR_ALIGN_THREADS=256 timeout -k 5 120 python scripts/pair_probe.py 4 rep6 2>&1 | tail -2
  SEQRUSH_AMD_LIB=$GRAFT_REPO_ROOT/seqrush_amd/libseqrush_amd_$v.so SR_NWG=1 SR_BLK_LEVELS=5 SR_ALIGN_THREADS=256 timeout -k 5 120 python scripts/pair_probe.py 4 all 2>&1 | tail -1
done
