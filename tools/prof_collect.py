#!/usr/bin/env python3
"""Condense a tools/prof_pmc.sh output directory (gpurun_out/prof_<tag>) into the two small files that
are committed under profiles/:

    profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, urh:: kernels only
                                       (torch's generator kernels have multi-KB names and are dropped)
    profiles/<name>_pmc.json           per-kernel mean PMC counters + derived HBM traffic per launch

usage: tools/prof_collect.py gpurun_out/prof_<tag> profiles/<name>

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE
are collected in separate passes, are reported in KiB, and on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled for kernels whose reads are
all of that shape (k_demod_runs, k_fir, k_costas*); WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = []
    for f in sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "urh::" in r["Name"]:
                stats.append(r)
    with open(dst + "_kernel_stats.csv", "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in stats:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    res = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "urh::" in r["Kernel_Name"] and r["Kernel_Name"] not in res:
                res[r["Kernel_Name"]]["resources"] = {k: int(r[k]) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                                            "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k].setdefault("pmc_mean", {})[c] = sum(v) / len(v)
                res[k].setdefault("pmc_launches", {})[c] = len(v)
    for r in stats:
        res[r["Name"]]["avg_ns"] = float(r["AverageNs"])
        res[r["Name"]]["calls"] = int(r["Calls"])
    for k, d in res.items():
        pm = d.get("pmc_mean", {})
        if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
            d["hbm_traffic_bytes_per_launch"] = {
                "read": pm["FETCH_SIZE"] * 1024 * 2, "write": pm["WRITE_SIZE"] * 1024,
                "total": pm["FETCH_SIZE"] * 1024 * 2 + pm["WRITE_SIZE"] * 1024,
                "note": "FETCH_SIZE[KiB] x 2 (gfx950 wide-read correction) + WRITE_SIZE[KiB]"}
    json.dump(res, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)
    for k, d in res.items():
        print(k[:80], d.get("avg_ns"), d.get("hbm_traffic_bytes_per_launch", {}).get("total"))


if __name__ == "__main__":
    main()
