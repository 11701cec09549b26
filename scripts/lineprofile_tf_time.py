#!/usr/bin/env python3
"""End-to-end time of the reference's DEFAULT line profile -- lineprofile(m, x, d) = TransferFunctionMethod, numrₑ = 100, N = 80
samples per radius ("a handful of seconds" on the reference's CPU path, docs/src/lineprofiles.md:66) -- on the device
tracer, with the wall time split into kernel launches and host work.

    python scripts/lineprofile_tf_time.py [root_finder=default|reference]"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import gradus_jl_amd as G
from gradus_jl_amd import transfer_functions as TF

ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(60), 0.0])        # the documentation's example geometry
d = G.ThinDisc(0.0, float("inf"))
bins = np.linspace(0.1, 1.5, 180)
kw = {}
if len(sys.argv) > 1 and sys.argv[1] == "reference":
    kw["root_finder"] = "reference"
out = {}
for rep in range(3):
    TF.LAUNCH_LOG = []
    t0 = time.perf_counter()
    b, f = G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, x, d, maxrₑ=50.0, ensemble=ens, **kw)
    wall = time.perf_counter() - t0
    log = TF.LAUNCH_LOG
    TF.LAUNCH_LOG = None
    by = {}
    for name, n, k, c in log:
        e = by.setdefault(name, [0, 0, 0.0, 0.0])
        e[0] += 1; e[1] += n; e[2] += k; e[3] += c
    out = {"wall_s": wall, "launches": len(log), "rays": sum(n for _, n, _, _ in log),
           "kernel_ms_total": sum(k for _, _, k, _ in log), "call_ms_total": sum(c for _, _, _, c in log),
           "host_s": wall - sum(c for _, _, _, c in log) / 1e3,
           "by_entry_point": {k: {"launches": v[0], "rays": v[1], "kernel_ms": v[2], "call_ms": v[3]} for k, v in by.items()},
           "flux_sum": float(np.nansum(f)), "peak_g": float(b[int(np.nanargmax(f))])}
    print(f"run {rep}: wall {wall:.3f} s, {len(log)} launches, {out['rays']} rays, kernels {out['kernel_ms_total']:.1f} ms, "
          f"device calls {out['call_ms_total']:.1f} ms, host {out['host_s']:.3f} s", flush=True)
print(json.dumps(out))
