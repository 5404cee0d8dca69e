#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of scripts/profile_round.sh: per launch of the alignment / orientation / unite
kernels -- average duration (kernel-trace stats), HBM bytes (FETCH_SIZE x calibrated factor + WRITE_SIZE x factor,
scripts/calib), SQ instruction and cycle counters, VALU issue fraction."""
import csv
import glob
import json
import os
import sys

FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0       # profiles/r02_calibration.json: 8 B/lane, 512-B chunks: 2.002 / 1.000
VALU_PEAK_PER_SIMD = 0.5                    # wave-instructions per cycle and SIMD (calibration: 0.44 at 8 waves/SIMD)


def kname(n):
    if "sr_align_blk_kernel" in n or "sr_align_bfs_kernel" in n:
        return "align"
    if "sr_orient_kernel" in n or "sr_orient_blk_kernel" in n:
        return "orient"
    if "sr_unite_kernel" in n:
        return "unite"
    return None


def counters(d):
    """kind -> counter -> mean per dispatch (summed over the dimensions rocprofv3 splits a counter into)"""
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = kname(row["Kernel_Name"])
                if not k:
                    continue
                key = (k, row["Counter_Name"], row["Dispatch_Id"])
                per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    out = {}
    for (k, c, _), v in per.items():
        out.setdefault(k, {}).setdefault(c, []).append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


def main():
    d, tag = sys.argv[1], sys.argv[2]
    res = {"round_tag": tag, "command": "rocprofv3 --pmc <one group per pass> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 "
                                        "--no-cpu-baseline; durations from the --kernel-trace --stats pass (--steps 3 --warmup 1)",
           "fetch_factor": FETCH_FACTOR, "write_factor": WRITE_FACTOR, "valu_peak_insts_per_cycle_per_simd": VALU_PEAK_PER_SIMD}
    for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = kname(row["Name"])
                if k:
                    res[f"{k}_kernel"] = row["Name"]
                    res[f"{k}_kernel_ms"] = float(row["AverageNs"]) / 1e6
                    res[f"{k}_calls"] = int(row["Calls"])
    fetch = counters(os.path.join(d, "fetch")); write = counters(os.path.join(d, "write"))
    sq = counters(os.path.join(d, "sq1"))
    for sub in ("sq2", "tcc1", "tcc2", "sq3", "sq4"):
        for k, v in counters(os.path.join(d, sub)).items():
            sq.setdefault(k, {}).update(v)
    res["host"] = os.environ.get("SR_PROFILE_HOST", "unknown")
    res["git"] = os.environ.get("SR_PROFILE_GIT", "unknown")
    for k in ("align", "orient", "unite"):
        f = fetch.get(k, {}).get("FETCH_SIZE"); w = write.get(k, {}).get("WRITE_SIZE")
        if f is not None and w is not None:
            res[f"{k}_FETCH_SIZE_KiB"] = f; res[f"{k}_WRITE_SIZE_KiB"] = w
            res[f"{k}_hbm_bytes_per_launch"] = f * 1024 * FETCH_FACTOR + w * 1024 * WRITE_FACTOR
            res[f"{k}_hbm_bytes_per_launch_raw"] = (f + w) * 1024
            if res.get(f"{k}_kernel_ms"):
                res[f"{k}_hbm_GBps"] = res[f"{k}_hbm_bytes_per_launch"] / (res[f"{k}_kernel_ms"] * 1e-3) / 1e9
        for c, v in sq.get(k, {}).items():
            res[f"{k}:{c}"] = v
        rd, wr = sq.get(k, {}).get("TCC_EA0_RDREQ_DRAM_32B_sum"), sq.get(k, {}).get("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum")
        if rd is not None and wr is not None:
            # exact byte counts of the requests the L2 sent towards DRAM (Infinity-Cache hits included: no counter separates them)
            res[f"{k}_dram_read_bytes_per_launch"] = rd * 32.0
            res[f"{k}_dram_write_bytes_per_launch"] = wr * 32.0
            res[f"{k}_dram_bytes_per_launch"] = (rd + wr) * 32.0
            res[f"{k}_mall_hit_bytes_per_launch"] = None
        hit, miss = sq.get(k, {}).get("TCC_HIT_sum"), sq.get(k, {}).get("TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            res[f"{k}_l2_hit_rate"] = hit / (hit + miss)
        wa, wc = sq.get(k, {}).get("SQ_WAIT_ANY"), sq.get(k, {}).get("SQ_WAVE_CYCLES")
        if wa and wc:
            res[f"{k}_wait_any_frac"] = wa / wc
        insts, gui = sq.get(k, {}).get("SQ_INSTS_VALU"), sq.get(k, {}).get("GRBM_GUI_ACTIVE")
        if insts:
            res[f"{k}_valu_insts"] = insts
            res[f"{k}_salu_insts"] = sq.get(k, {}).get("SQ_INSTS_SALU")
        ireq, imiss = sq.get(k, {}).get("SQC_ICACHE_REQ"), sq.get(k, {}).get("SQC_ICACHE_MISSES")
        if ireq and imiss is not None:
            res[f"{k}_icache_miss_rate"] = imiss / ireq
        if insts and gui:
            cycles = gui / 8.0                                   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            res[f"{k}_busy_cycles"] = cycles
            res[f"{k}_valu_insts_per_cycle_per_simd"] = insts / (1024.0 * cycles)
            res[f"{k}_valu_issue_frac"] = insts / (1024.0 * cycles) / VALU_PEAK_PER_SIMD
    try:
        res["bench_plain"] = json.loads(open(os.path.join(d, "bench_plain.json")).read().strip().split("\n")[-1])
        res["bench_under_stats"] = json.loads(open(os.path.join(d, "bench_under_stats.json")).read().strip().split("\n")[-1])
    except Exception as e:
        res["bench_error"] = str(e)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
