#!/usr/bin/env python3
"""Wall time of the blocking end-point calls (152 B per ray back to the host) against their kernel time: prerendergeodesics at
2048² (637 MB) and tracegeodesics on 1 M (x, v) pairs."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)

import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
d = G.ThinDisc(m.isco(), 50.0)
out = {}
for S in (1024, 2048):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        a, b, cache = G.prerendergeodesics(m, x, d, 2000.0, image_width=S, image_height=S, alpha_lims=(-60, 60), beta_lims=(-35, 35),
                                           ensemble=ens)
        ts.append(time.perf_counter() - t0)
    for pipe in (4, 0):
        ens.set("pipeline", pipe)
        tt = []
        for _ in range(5):
            cfg0 = G.render_configuration(m, x, d, 2000.0, image_width=S, image_height=S, alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens)
            t0 = time.perf_counter()
            pts0, st0 = G.ensemble_solve_tracing_problem(ens, cfg0, stats=True)
            tt.append(time.perf_counter() - t0)
            del pts0
        out[f"endpoints_{S}_pipeline{pipe}_ms"] = [round(t * 1e3, 2) for t in tt] + [f"kernels {st0['kernel_ms']:.2f}"]
    ens.set("pipeline", 4)
    cfg = G.render_configuration(m, x, d, 2000.0, image_width=S, image_height=S, alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens)
    tr = []
    for _ in range(4):
        t0 = time.perf_counter()
        pts = G.ensemble_solve_tracing_problem(ens, cfg)
        tr.append(time.perf_counter() - t0)
    out[f"prerender_{S}"] = {"wall_ms": [round(t * 1e3, 2) for t in ts], "gr_render_endpoints_ms": [round(t * 1e3, 2) for t in tr],
                             "MB": S * S * 152 / 1e6}
print(json.dumps(out))
