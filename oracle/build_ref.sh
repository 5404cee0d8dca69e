#!/bin/bash
# ORACLE tooling: build the WFA v1 driver from the crate that ships in the
# reference's cargo cache.  The crate is unpacked and compiled in a scratch
# directory OUTSIDE the repo; only the resulting binary lands in oracle/_ref/
# (git-ignored).  Needs /root/reference, so it only runs in the build container.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
CRATE=/root/reference/.cargo_home/registry/cache/index.crates.io-1949cf8c6b5b557f/libwfa-0.1.2.crate
[ -f "$CRATE" ] || { echo "build_ref: reference crate not present, skipping"; exit 0; }
OUT="$HERE/_ref"
mkdir -p "$OUT"
if [ -x "$OUT/wfa1_driver" ] && [ "$OUT/wfa1_driver" -nt "$HERE/ref_wfa1_driver.c" ]; then exit 0; fi
SCRATCH="${TMPDIR:-/tmp}/sr_ref_build"
rm -rf "$SCRATCH"; mkdir -p "$SCRATCH"
tar xzf "$CRATE" -C "$SCRATCH"
W="$SCRATCH/libwfa-0.1.2/WFA"
make -C "$W" -s clean all >/dev/null 2>&1 || make -C "$W" -s all >/dev/null 2>&1
gcc -O2 -I"$W" "$HERE/ref_wfa1_driver.c" "$W"/build/*.o -lm -lrt -o "$OUT/wfa1_driver"
echo "build_ref: built $OUT/wfa1_driver"
