#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 counter passes tools/scripts/train_pmc.sh (or any script with the same layout) leaves
under a directory: <dir>/<prefix>_FETCH_SIZE, <prefix>_WRITE_SIZE, <prefix>_SQ_VALU_MFMA_BUSY_CYCLES (each a rocprofv3 -d tree
with *counter_collection.csv).  Per launch: read = FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md: the counter
under-counts reads by half), write = WRITE_SIZE; rocprofv3 reports both derived metrics in KB.

    python tools/pmc_summary.py gpurun_out/r2x train "command line of the profiled run" > profiles/rNN_..._pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    """kernel name -> counter name -> list of per-launch values (summed over dimensions of one dispatch)."""
    out = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k, c, disp = r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"]
            out[k][c][disp] = out[k][c].get(disp, 0.0) + float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out, dur


def main():
    root, prefix, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, dur = load(os.path.join(root, prefix + "_FETCH_SIZE"))
    write, _ = load(os.path.join(root, prefix + "_WRITE_SIZE"))
    busy, _ = load(os.path.join(root, prefix + "_SQ_VALU_MFMA_BUSY_CYCLES"))
    total = sum(sum(v) for v in dur.values())
    res = {}
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if sum(dur[k]) < 0.01 * total:
            continue
        mean = lambda m: (sum(m.values()) / len(m)) if m else None
        rd, wr = mean(fetch[k].get("FETCH_SIZE", {})), mean(write[k].get("WRITE_SIZE", {}))
        b, g = mean(busy[k].get("SQ_VALU_MFMA_BUSY_CYCLES", {})), mean(busy[k].get("GRBM_GUI_ACTIVE", {}))
        us = sum(dur[k]) / len(dur[k])
        e = {"launches": len(dur[k]), "avg_us_under_pmc": us}
        if rd is not None and wr is not None:
            e["read_MB_x2"] = rd * 1024.0 / 1e6 * 2.0       # the counters are reported in KB
            e["write_MB"] = wr * 1024.0 / 1e6
            e["traffic_TBps"] = (e["read_MB_x2"] + e["write_MB"]) / us     # MB / us = TB/s
        if b is not None and g:
            e["mfma_busy"] = b / (g / 8.0 * 1024.0)
        res[k[:60]] = e
    print(json.dumps({"command": cmd, "note": "per-launch means; read = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md), "
                      "write = WRITE_SIZE; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024); kernels above 1 % "
                      "of the run's kernel time", "kernels": res}, indent=1))


if __name__ == "__main__":
    main()
