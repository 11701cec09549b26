#!/usr/bin/env python3
"""Wall time of the corona -> disc calls against the trace they contain (host-side sampler, tetrad and reductions are numpy)."""
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
d = G.ThinDisc(0.0, 1000.0)
model = G.LampPostModel(h=10.0)
out = {}
for n in (10_000, 100_000, 1_000_000):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        cg = G.tracecorona(m, d, model, n_samples=n, ensemble=ens)
        ts.append(time.perf_counter() - t0)
    out[f"tracecorona_{n}"] = [round(t * 1e3, 1) for t in ts]
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        prof = G.emissivity_profile(m, d, model, n_samples=n, ensemble=ens, sampler=G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator()))
        ts.append(time.perf_counter() - t0)
    out[f"emissivity_profile_mc_{n}"] = [round(t * 1e3, 1) for t in ts]
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    prof = G.emissivity_profile(m, d, model, n_samples=1000, ensemble=ens)
    ts.append(time.perf_counter() - t0)
out["emissivity_profile_point_source_1000"] = [round(t * 1e3, 1) for t in ts]
print(json.dumps(out))
