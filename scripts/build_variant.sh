#!/bin/bash
# A/B build: scripts/build_variant.sh NAME "-DFLAG=.. ..." -> seqrush_amd/libseqrush_amd_NAME.so (only the 2-bit blocked
# alignment unit is recompiled with the extra flags; select it at run time with SEQRUSH_AMD_LIB=<path>)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../seqrush_amd/csrc"
mkdir -p build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSR_SYMBITS=2 -DSR_BUILD_TAG="\"$name\"" $flags -c -o build_$name/sr_align_blk_s2.o sr_align_blk.hip
objs=$(ls build/*.o | grep -v "build/sr_align_blk_s2.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libseqrush_amd_$name.so $objs build_$name/sr_align_blk_s2.o
# (build() goes by the digest of the sources written next to the default library, not by file times: a variant build leaves
# the default artefacts alone; the workspace report's kernel_build / source_digest name what ran)
echo built ../libseqrush_amd_$name.so
