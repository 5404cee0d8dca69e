#!/usr/bin/env python3
"""Generates tests/golden/wfa1_vectors.json by RUNNING the WFA v1 C code that
ships in the reference's cargo cache (oracle/build_ref.sh -> oracle/_ref/wfa1_driver).
Run in the build container only (needs /root/reference):

    bash oracle/build_ref.sh && python tests/golden/gen_wfa1_vectors.py

The fixture holds inputs and outputs only (score + raw CIGAR), no source text.
WFA v1 is single-piece gap-affine; its optimal SCORE must equal WFA2's and this
repo's oracle; its CIGAR tie-breaking (del-ext, del-open, ins-ext, ins-open,
mismatch) is NOT WFA2's, so CIGARs are asserted only for the reference's own
boundary inputs (tests/test_wfa2_cigar_debug.rs, tests/test_cigar_validity.rs).
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from seqrush_amd import synth  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "wfa1_driver")


def main():
    cases = []
    # the reference's own WFA boundary inputs (affine 0,5,8,2)
    ref_inputs = [
        ("ATCGATCG", "ATCGATCGATCG", "tests/test_wfa2_cigar_debug.rs:11-29"),
        ("ATCGATCGATCG", "ATCGATCGATCG", "tests/test_cigar_validity.rs:9-13"),
        ("ATCGATCGATCG", "ATCGATCGAT", "tests/test_cigar_validity.rs:14-18"),
        ("ATCGATCGAT", "ATCGATCGATCG", "tests/test_cigar_validity.rs:19-23"),
        ("ATCGATCGATCG", "ATTGATCGATCG", "tests/test_cigar_validity.rs:24-28"),
    ]
    for p, t, src in ref_inputs:
        cases.append(dict(pattern=p, text=t, x=5, o=8, e=2, source=src, assert_cigar=True))
    # synthetic families (seeded), several penalty sets
    k = 0
    for L, sub, indel in [(60, 0.05, 0.02), (150, 0.1, 0.03), (400, 0.05, 0.01), (400, 0.2, 0.05),
                          (1200, 0.05, 0.01)]:
        fam = synth.indel_family(3, L, sub, indel, 9000 + k)
        k += 1
        for (x, o, e) in [(5, 8, 2), (1, 1, 1), (4, 6, 2)]:   # WFA v1 rejects gap_open = 0
            for a in range(3):
                for b in range(3):
                    if a == b:
                        continue
                    cases.append(dict(pattern=fam[a][1].decode(), text=fam[b][1].decode(), x=x, o=o, e=e,
                                      source=f"synth.indel_family(3,{L},{sub},{indel},{9000 + k - 1})[{a}] vs [{b}]",
                                      assert_cigar=False))
    inp = "".join(f"{c['pattern']} {c['text']} {c['x']} {c['o']} {c['e']}\n" for c in cases)
    out = subprocess.run([DRIVER], input=inp.encode(), stdout=subprocess.PIPE, check=True, timeout=300).stdout.decode()
    lines = out.strip().split("\n")
    assert len(lines) == len(cases)
    for c, ln in zip(cases, lines):
        sc, cig = ln.split(" ")
        c["score"] = int(sc)
        c["cigar"] = cig
        if not c["assert_cigar"] and len(c["pattern"]) > 200:
            # keep the fixture small: long inputs are regenerated from the seed
            c.pop("pattern"); c.pop("text"); c.pop("cigar")
    json.dump(dict(generator="tests/golden/gen_wfa1_vectors.py", reference="WFA v1 (libwfa-0.1.2.crate, cargo cache of the reference)",
                   cases=cases), open(os.path.join(HERE, "wfa1_vectors.json"), "w"), indent=0)
    print(f"wrote {len(cases)} cases")


if __name__ == "__main__":
    main()
