#!/usr/bin/env python3
"""Fold a `rocprofv3 --pmc ... -- scripts/microbench/valu_calib` run into the calibration table of the utilisation
numbers (profiles/r3_valu_calib.json): per instruction class the sustained shader clock, SIMD cycles per
wave-instruction, and what the `valu_busy` formula of scripts/summarize_pmc.py reads on a stream that keeps the VALU
busy by construction.

    python scripts/summarize_calib.py <rocprof out dir> [bare-run jsonl]

Definitions (one launch, 8 XCDs, 1024 SIMDs):
    clock_ghz      = GRBM_GUI_ACTIVE / 8 / duration                 (GRBM_GUI_ACTIVE is summed over the XCDs)
    simd_cycles    = GRBM_GUI_ACTIVE / 8 * 1024
    cyc_per_inst   = simd_cycles / SQ_INSTS_VALU
    valu_busy_raw  = 4 * SQ_ACTIVE_INST_VALU / simd_cycles          (the formula round 2 used: SQ_ACTIVE_INST_VALU in quad-cycles)
"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "calib_" not in name:
            continue
        name = name.split("(")[0].split()[-1]
        d = acc.setdefault(name, {}).setdefault(int(row["Dispatch_Id"]), {"dur_ns": int(row["End_Timestamp"]) - int(row["Start_Timestamp"])})
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
bare = {}
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    for line in open(sys.argv[2]):
        if line.startswith("{") and '"kernel"' in line:
            r = json.loads(line)
            bare.setdefault(r["kernel"], []).append(r)
res = {}
for name, disp in sorted(acc.items()):
    # the last launch of each class is the sized one (probe launches come first); take the longest
    d = max(disp.values(), key=lambda x: x["dur_ns"])
    g = d.get("GRBM_GUI_ACTIVE")
    e = {"dur_ms_under_pmc": d["dur_ns"] / 1e6}
    if g:
        simd_cycles = g / 8.0 * 1024.0
        e["clock_ghz"] = g / 8.0 / d["dur_ns"]
        if d.get("SQ_INSTS_VALU"):
            e["simd_cycles_per_valu_inst"] = simd_cycles / d["SQ_INSTS_VALU"]
        if d.get("SQ_ACTIVE_INST_VALU"):
            e["valu_busy_raw"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / simd_cycles
            if d.get("SQ_INSTS_VALU"):
                e["active_quads_per_inst"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]
        if d.get("SQ_BUSY_CYCLES"):
            e["sq_busy_over_gui_active"] = d["SQ_BUSY_CYCLES"] / g
    e["counters"] = {k: v for k, v in d.items() if k != "dur_ns"}
    if name in bare:
        best = max(bare[name], key=lambda r: r["wave_inst_per_s"])
        e["bare_wave_inst_per_s"] = best["wave_inst_per_s"]
        e["bare_ms"] = best["ms"]
        if "clock_ghz" in e:
            # cycles per instruction per SIMD from the bare rate at the clock the counters saw
            e["bare_simd_cycles_per_inst"] = e["clock_ghz"] * 1e9 * 1024.0 / best["wave_inst_per_s"]
    res[name] = e
json.dump(res, open(os.path.join(out, "calib_summary.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in res.items()}, indent=1))
