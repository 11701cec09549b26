"""debug: tangents through the table against the fused tangent kernel, ray by ray"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import gradus_jl_amd as G
from gradus_jl_amd.transfer_functions import device_tracer
ens = G.EnsembleMI355X(0)
kerr = G.KerrMetric(1.0, 0.998)
tab = G.TabulatedMetric(kerr)
x = np.array([0.0, 100_000.0, math.radians(30), 0.0])
rng = np.random.default_rng(2026)
rr, th = rng.uniform(2.5, 14.0, 300), rng.uniform(0.0, 2 * math.pi, 300)
th[:4] = [math.pi / 2, math.pi / 2 + 1e-3, 3 * math.pi / 2, 0.0]
al, be = rr * np.cos(th), rr * np.sin(th)
def tracer(m):
    return device_tracer(m, x, 2 * x[1], G.chart_for_metric(m, 2 * x[1], closest_approach=1.005), G.ConstPointFunctions.redshift(m, x, ensemble=ens), ens)
trt, trf = tracer(tab), tracer(kerr)
prev = None
for rep in range(8):
    ens.set("tangent_norm", 1 - rep % 2)
    trt, trf = tracer(tab), tracer(kerr)
    t, f = trt.tangent(al, be), trf.tangent(al, be)
    nt, nf = np.nonzero(~np.isfinite(t).all(axis=1))[0], np.nonzero(~np.isfinite(f).all(axis=1))[0]
    same = None if prev is None else (np.array_equal(t, prev[0], equal_nan=True), np.array_equal(f, prev[1], equal_nan=True))
    print("rep", rep, "nan rows table", nt.tolist()[:10], "fused", nf.tolist()[:10], "identical to previous rep (table, fused):", same)
    for i in nt[:3]:
        print("    ", i, al[i], be[i], t[i], f[i])
    prev = (t, f)
# bigger launch: one lane per ray
al2, be2 = np.tile(al, 300), np.tile(be, 300)
ens.set("tangent_pairs", 0)
import ctypes as C
t2 = trt.tangent(al2, be2)
print("90000 rays: nan rows", int((~np.isfinite(t2).all(axis=1)).sum()), "copies identical:", np.array_equal(t2[:300], t2[300:600], equal_nan=True), np.array_equal(t2[:300], t2[-300:], equal_nan=True))
