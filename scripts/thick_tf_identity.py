#!/usr/bin/env python3
"""Normalisation identity of the thick-disc transfer functions on the device (the two cases of test/transfer-functions/
test-thick-disc.jl): ∮ (f/g) 2 dφ (g✶ = sin²φ) against (1/π rₑ) dA/drₑ, A(rₑ) = area enclosed by the image of the ring ρ = rₑ on
the disc's surface (offsets found against datumplane(d, rₑ), A = ½∮ r(θ)² dθ about (α₀, β₀)); N = samples around the ring."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

import gradus_jl_amd as G
from gradus_jl_amd import transfer_functions as TF

ens = G.EnsembleMI355X(0)
for a, angle, r_e, edd, gold in ((0.998, 75, 3.0, 0.3, 14.64279128586961), (0.2, 20, 5.469668466100368, 0.2, 21.581370829241525)):
    m = G.KerrMetric(1.0, a)
    x = np.array([0.0, 10_000.0, math.radians(angle), 0.0])
    d = G.ShakuraSunyaev.for_metric(m, eddington_ratio=edd)
    chart = G.chart_for_metric(m, 2 * x[1])
    datum = TF.device_tracer(m, x, 2 * x[1], chart, G.ConstPointFunctions.redshift(m, x), ens)
    th = np.linspace(0.0, 2 * math.pi, 1441)[:-1]

    def area(r):
        h = float(d.cross_section(float(r)))
        rr = TF.find_offsets_for_radius(datum, np.full(th.size, r), th, r_min=m.inner_radius(), β0=2.0, heights=np.full(th.size, h))[0]
        return 0.5 * np.sum(rr * rr) * (th[1] - th[0])

    δ = 1e-3
    lo = max(r_e - δ, d.inner_radius + 1e-9)
    dA = (area(r_e + δ) - area(lo)) / (r_e + δ - lo)
    for N in (80, 400, 1600):
        c = G.cunningham_transfer_function(m, x, d, r_e, β0=2.0, ensemble=ens, N=N)
        gs = np.clip(c.g_star, 0.0, 1.0)
        y = c.f / (c.gmin + gs * (c.gmax - c.gmin))
        for k in np.flatnonzero(c.f == 0.0):
            y[k] = 0.5 * (y[k - 1] + y[(k + 1) % y.size])
        phi = np.arcsin(np.sqrt(gs))
        total = np.sum(0.5 * (y + np.roll(y, -1)) * 2.0 * np.abs(np.roll(phi, -1) - phi))
        print(f"a={a} θ={angle}° rₑ={r_e:.4f} N={N}: ∮(f/g)2dφ = {total:.5f}   dA/drₑ/(π rₑ) = {dA / (math.pi * r_e):.5f}   "
              f"ratio {total / (dA / (math.pi * r_e)):.4f}   Σf = {np.nansum(c.f):.5f}   finite {int(np.isfinite(c.f).sum())}/{c.f.size}")
