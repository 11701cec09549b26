#!/usr/bin/env python3
"""Summarise rocprofv3 output of scripts/profile_pmc.sh into one JSON -- the file that gets committed under profiles/
and that bench.py reads to price a launch.

    python scripts/summarize_pmc.py <out_dir> [kernel-name-substring] [warmup-launches-to-drop]

Per-launch means of the DOMINANT kernel whose name contains the substring (default "k_trace"): durations from the
kernel trace with the first `warmup` launches dropped (the cold launch is not a timed launch), counters from the
--pmc passes (each pass is its own run: the first `warmup` dispatches are dropped there too), FETCH_SIZE doubled as
MI355X_MICROARCH.md §HBM prescribes for gfx950.  `executed` is the FP64 work the kernel really issued:
(2 FMA + MUL + ADD + TRANS) wave-instructions x 64 lanes x active-lane fraction.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

out = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
res = {"kernel": None, "counters": {}, "warmup_launches_dropped": warm}

try:
    from gradus_jl_amd._lib import kernel_source_sha16
    res["source_sha16"] = kernel_source_sha16()
except Exception as e:      # noqa: BLE001
    res["source_sha16"] = None
    res["source_sha16_error"] = str(e)

# ---- durations ----
durs = {}
meta = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if needle not in name:
            continue
        durs.setdefault(name, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        meta.setdefault(name, row)
if durs:
    name = max(durs, key=lambda k: sum(d for _, d in durs[k]))
    seq = [d for _, d in sorted(durs[name])]
    timed = seq[warm:] if len(seq) > warm else seq
    row = meta[name]
    res.update({"kernel": name, "calls": len(seq), "timed_calls": len(timed),
                "avg_ms": sum(timed) / len(timed) / 1e6, "min_ms": min(timed) / 1e6, "max_ms": max(timed) / 1e6,
                "cold_ms": seq[0] / 1e6,
                # rocprofv3 reports VGPR_Count in units of a wave32 allocation: the registers per LANE of a wave64 are twice that
                "arch_vgpr_rocprof_units": int(row["VGPR_Count"]), "arch_vgpr_per_lane": 2 * int(row["VGPR_Count"]),
                "accum_vgpr_per_lane": 2 * int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                "scratch_bytes": int(row.get("Scratch_Size", 0) or 0), "lds_bytes": int(row.get("LDS_Block_Size", 0) or 0),
                "grid": int(row["Grid_Size_X"]), "workgroup": int(row["Workgroup_Size_X"])})

# ---- counters ----
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        if res["kernel"] and row["Kernel_Name"] != res["kernel"]:
            continue
        if not res["kernel"] and needle not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(int(row["Dispatch_Id"]), 0.0)
        acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for k, per_dispatch in acc.items():
        vals = [per_dispatch[d] for d in sorted(per_dispatch)]
        vals = vals[warm:] if len(vals) > warm else vals
        res["counters"][k] = sum(vals) / len(vals)
c = res["counters"]
if c.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in c:
    res["valu_lane_utilization"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
    res["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
if c.get("GRBM_GUI_ACTIVE") and res.get("avg_ms"):
    # sustained shader clock of the launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (res["avg_ms"] * 1e6)
if c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    res["simd_cycles_per_launch"] = simd_cycles
    # Calibrated on scripts/microbench/valu_calib.hip (profiles/r3_valu_calib.json): SQ_ACTIVE_INST_VALU counts ONE quad-cycle
    # per ordinary VALU instruction and four per transcendental, whatever the instruction really takes -- a pure
    # v_mov_b32 stream reads 1.10 here, v_fma_f32 0.99, v_fma/mul/add_f64 0.76-0.77 (5.2 clocks each, not 4), v_rcp_f64 0.97.
    # So this is "VALU instructions issued per 4 clocks", NOT a utilisation: it can exceed 1 (round 2 called it valu_busy).
    res["valu_issue_per_4clk"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
    if c.get("SQ_INSTS_VALU"):
        res["simd_cycles_per_valu_inst"] = simd_cycles / c["SQ_INSTS_VALU"]
if "FETCH_SIZE" in c:
    res["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0   # gfx950: FETCH_SIZE reads 1/2 (MI355X_MICROARCH.md §HBM)
if "WRITE_SIZE" in c:
    res["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0
f64 = [c.get(k) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")]
if all(v is not None for v in f64) and "valu_lane_utilization" in res:
    res["executed_fp64_flops_per_launch"] = (2.0 * f64[0] + f64[1] + f64[2] + f64[3]) * 64.0 * res["valu_lane_utilization"]
    if "simd_cycles_per_launch" in res:
        # the utilisation figure that cannot exceed 1 by construction: cycles the FP64 pipe is held at its NOMINAL cost
        # (4 clocks per FMA/MUL/ADD wave-instruction, 16 per transcendental) over the cycles available
        res["fp64_pipe_busy_nominal"] = (4.0 * (f64[0] + f64[1] + f64[2]) + 16.0 * f64[3]) / res["simd_cycles_per_launch"]
        res["fp64_inst_share_of_valu"] = sum(f64) / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None
f32 = [c.get(k) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32")]
if all(v is not None for v in f32) and "valu_lane_utilization" in res:
    res["executed_fp32_flops_per_launch"] = (2.0 * f32[0] + f32[1] + f32[2] + f32[3]) * 64.0 * res["valu_lane_utilization"]
bl = os.path.join(out, "bench.log")
if os.path.exists(bl):
    for line in open(bl):
        if line.startswith("{"):
            try:
                res["bench"] = json.loads(line)
            except Exception:      # noqa: BLE001
                pass
if "bench" in res and isinstance(res["bench"], dict):
    rays = res["bench"].get("config", {}).get("rays_per_gpu") or res["bench"].get("rays")
    if rays:
        res["rays_per_launch"] = rays
        if "executed_fp64_flops_per_launch" in res:
            res["executed_fp64_flops_per_ray"] = res["executed_fp64_flops_per_launch"] / rays
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "bench"}, indent=1))
