#!/usr/bin/env python3
"""Digest of the sources libseqrush_amd.so is built from: sha256 over (name, content) of seqrush_amd/csrc/*.{hip,inc,h,cpp},
its Makefile and include/seqrush_amd.h, first 16 hex digits.  The Makefile compiles it into the library
(srk_source_digest(), "source_digest" of the workspace report) and writes it next to the library
(libseqrush_amd.so.digest); __graft_entry__.build() rebuilds when the digest of the tree differs from that file --
mtimes say nothing after a checkout, a copy to the GPU box or a variant build."""
import hashlib
import os
import sys


def digest(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "seqrush_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))
             if f.endswith((".hip", ".inc", ".h", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(root, "include", "seqrush_amd.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(digest())
