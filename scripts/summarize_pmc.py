#!/usr/bin/env python3
"""Summarise rocprofv3 output of scripts/profile_pmc.sh into one JSON (per-launch means of the
trace kernel) -- the file that gets committed under profiles/."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
res = {"kernel": None, "counters": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Name"]:
            res["kernel"] = row["Name"]
            res["calls"] = int(row["Calls"])
            res["avg_ms"] = float(row["AverageNs"]) / 1e6
            res["min_ms"] = float(row["MinNs"]) / 1e6
            res["max_ms"] = float(row["MaxNs"]) / 1e6
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Kernel_Name"]:
            res["vgpr"] = int(row["VGPR_Count"]); res["agpr"] = int(row["Accum_VGPR_Count"]); res["sgpr"] = int(row["SGPR_Count"])
            res["grid"] = int(row["Grid_Size_X"]); res["workgroup"] = int(row["Workgroup_Size_X"])
            break
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "k_trace" not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res["counters"][k] = sum(v) / len(v)
c = res["counters"]
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
    # active lanes per VALU issue cycle (of 64)
    res["valu_lane_utilization"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) / 4.0 * 4.0
if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
    res["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
if "FETCH_SIZE" in c:
    res["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0   # gfx950: FETCH_SIZE reads 1/2 (MI355X_MICROARCH.md §HBM)
if "WRITE_SIZE" in c:
    res["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0
bl = os.path.join(out, "bench.log")
if os.path.exists(bl):
    for line in open(bl):
        if line.startswith("{"):
            res["bench"] = json.loads(line)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "bench"}, indent=1))
