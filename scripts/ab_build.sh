#!/bin/bash
# builds experiments/tmp/libA.so from HEAD and experiments/tmp/libB.so from the working tree
set -e
mkdir -p "$(dirname "$0")/../experiments/tmp"
cd "$(dirname "$0")/../seqrush_amd/csrc"
make -j8 -s 2>&1 | grep -E "error" && exit 1
cp ../libseqrush_amd.so ../../experiments/tmp/libB.so
git stash -q
make -j8 -s 2>&1 | grep -E "error" || true
cp ../libseqrush_amd.so ../../experiments/tmp/libA.so
git stash pop -q
make -j8 -s 2>&1 | grep -E "error" || true
echo built A=HEAD B=worktree
