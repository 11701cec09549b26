#!/usr/bin/env python3
"""Refresh the numeric cells of DESIGN.md §5's kernel tables from profiles/<name>_summary.json.

A table row is recognised by its last cell, `` `r5z_<workload>` ``; its cells 3..9 (ms per launch, M rays/s, executed TFLOP/s,
frac of peak, VALU issue per 4 clk, lane utilisation, VGPR / scratch) are rewritten from that summary.  Labels and workload
descriptions (cells 1, 2) are left alone.      python scripts/design_tables.py [--check]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK64, PEAK32 = 78.6, 157.3


def cells(name):
    d = json.load(open(os.path.join(ROOT, "profiles", f"{name}_summary.json")))
    ms = d["avg_ms"]
    rays = d.get("rays_per_launch") or (d.get("bench") or {}).get("rays")
    f32 = "executed_fp32_flops_per_launch" in d and "executed_fp64_flops_per_launch" not in d
    flops = d.get("executed_fp32_flops_per_launch" if f32 else "executed_fp64_flops_per_launch")
    tf = flops / (ms * 1e-3) / 1e12 if flops else None
    vg = d.get("arch_vgpr_per_lane", 0) + d.get("accum_vgpr_per_lane", 0)
    mr = rays / ms / 1e3 if rays else None
    fmt = lambda v, p: "–" if v is None else f"{v:.{p}f}"
    return [f"{ms:.2f}", fmt(mr, 0), fmt(tf, 1), fmt(tf / (PEAK32 if f32 else PEAK64) if tf else None, 2),
            f"{d['valu_issue_per_4clk']:.2f}", f"{d['valu_lane_utilization']:.2f}", f"{vg} / {d.get('scratch_bytes', 0)}"], d["source_sha16"]


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    lines = open(path).read().split("\n")
    hashes, changed = set(), 0
    for i, ln in enumerate(lines):
        m = re.search(r"\| `(r\d\w*_\w+)` \|\s*$", ln)
        if not m or not ln.startswith("|"):
            continue
        parts = ln.split("|")
        if len(parts) != 12:          # leading '' + 10 cells + trailing ''
            continue
        new, h = cells(m.group(1))
        hashes.add(h)
        old = [p.strip() for p in parts[3:10]]
        if old != new:
            changed += 1
            parts[3:10] = [f" {c} " for c in new]
            lines[i] = "|".join(parts)
    print(f"{changed} rows refreshed; source hashes of the summaries: {sorted(hashes)}")
    if "--check" in sys.argv:
        sys.exit(1 if changed else 0)
    open(path, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
