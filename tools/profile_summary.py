#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh into one JSON: per-kernel duration statistics from the
kernel-trace pass and per-dispatch means of every PMC counter, for the libttrnn kernels only.

Units (MI355X guide, HBM/rocprofv3 section): FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports HALF of
wide coalesced reads, so HBM read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.  SQ_* are quad-cycles except
SQ_VALU_MFMA_BUSY_CYCLES (cycles)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """kernel base name + the template arguments that tell instantiations apart (shape digits, flags)"""
    m = re.search(r"(k_[a-z0-9_]+)(<.*>)?", name)
    if not m:
        return name[:60]
    args = m.group(2) or ""
    shp = re.search(r"Shp<([0-9, ]+)>", args)
    tag = ""
    if shp:
        d = [int(v) for v in shp.group(1).split(",")]
        tag = "[J=%s I=%s R=%s]" % ("x".join(map(str, d[1:1 + d[0]])), "x".join(map(str, d[5:5 + d[0]])),
                                    "x".join(map(str, d[9:9 + d[0] - 1])))
        rest = re.sub(r"ttrnn::Shp<[0-9, ]+>,?\s*", "", args)
        rest = rest.strip("<> ")
        if rest:
            tag += "<" + rest + ">"
    return m.group(1) + tag


def main(out):
    res = {"kernel_stats": {}, "pmc_mean_per_dispatch": {}}
    for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "ttrnn" not in row.get("Name", ""):
                    continue
                res["kernel_stats"][short(row["Name"])] = {
                    "calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                    "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3,
                    "total_ms": float(row["TotalDurationNs"]) / 1e6, "pct": float(row["Percentage"])}
    for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "ttrnn" not in row.get("Kernel_Name", ""):
                    continue
                key = (short(row["Kernel_Name"]), row["Counter_Name"])
                tot, n = acc.get(key, (0.0, 0))
                acc[key] = (tot + float(row["Counter_Value"]), n + 1)
        for (kern, ctr), (tot, n) in acc.items():
            res["pmc_mean_per_dispatch"].setdefault(kern, {})[ctr] = tot / n
            res["pmc_mean_per_dispatch"][kern].setdefault("_dispatches", n)
    for kern, c in res["pmc_mean_per_dispatch"].items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes_per_dispatch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
    res["notes"] = __doc__
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
