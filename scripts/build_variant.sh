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
# (the GPU box's build() runs make when a source is newer than libseqrush_amd.so -- and then the "default" of an A/B is
# the working tree too: keep the libraries newer than the sources; the workspace report's kernel_build names what ran)
touch ../libseqrush_amd*.so ../seqrush_mi355x 2>/dev/null || true
echo built ../libseqrush_amd_$name.so
