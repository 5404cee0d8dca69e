#!/usr/bin/env python3
"""Merge calib's known byte counts with the rocprofv3 FETCH_SIZE / WRITE_SIZE passes -> correction factors
(true bytes / counter bytes) per access shape, and the measured VALU issue rate."""
import csv
import glob
import json
import os
import sys


def counter_sums(d, counter):
    """kernel name -> list of per-dispatch counter values (KiB on gfx950)"""
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                out.setdefault(row["Kernel_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                out[row["Kernel_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: list(v.values()) for k, v in out.items()}


def main():
    d = sys.argv[1]
    plain = json.load(open(os.path.join(d, "plain.json")))
    fetch = counter_sums(os.path.join(d, "fetch"), "FETCH_SIZE")
    write = counter_sums(os.path.join(d, "write"), "WRITE_SIZE")
    res = {"_note": "factor = known bytes / (counter KiB x 1024); multiply a kernel's FETCH_SIZE / WRITE_SIZE by the factor "
                    "of its access shape.  sr_align_blk_kernel moves its rows as 8 B/lane, 512-B chunks (the *_scatter rows).",
           "hbm": [], "valu": plain["valu"]}
    for k in plain["hbm"]:
        name = k["kernel"]
        row = dict(k)
        for label, tab in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            vals = [v for kn, vs in tab.items() if name.split("<")[0] in kn and
                    (("<" not in name) or (("true" in name) == ("true" in kn or "1>" in kn))) for v in vs]
            if vals:
                kib = sum(vals) / len(vals)
                row[label + "_KiB_per_dispatch"] = kib
                row[label + "_factor"] = (k["bytes"] / (kib * 1024.0)) if kib > 0 else None
        res["hbm"].append(row)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
