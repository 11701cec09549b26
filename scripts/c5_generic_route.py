#!/usr/bin/env python3
"""BASELINE C5's plane with the three kinds of emissivity: a power law (fused), an emissivity profile = a table of radii
(fused: interpolated on the device), an arbitrary callable (generic route: (g, ρ) pairs back to the host, ε and the bucketing
in numpy)."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
u = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(m.isco(), 250.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plane = G.PolarPlane(G.GeometricGrid(), Nr=N, Nθ=N, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)
out = {}
from gradus_jl_amd.corona import RadialDiscProfile

rr = np.geomspace(m.isco(), 250.0, 100)
table = RadialDiscProfile(rr, rr ** -3.0, np.zeros_like(rr))            # what emissivity_profile(...) returns, here a sampled power law
for name, eps in (("power_law_fused", G.PowerLawEmissivity(3)), ("emissivity_profile_fused", table), ("callable_generic", lambda r: r ** -3.0)):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        xs, ys, st = G.lineprofile(bins, eps, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0, ensemble=ens, stats=True)
        ts.append(time.perf_counter() - t0)
    out[name] = {"wall_ms": [round(t * 1e3, 1) for t in ts], "kernel_ms": round(st["kernel_ms"], 1), "y90": float(ys[90])}
print(json.dumps(out))
