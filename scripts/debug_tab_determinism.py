"""debug: is the tangent kernel of a tabulated metric deterministic over repeated small launches?"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import gradus_jl_amd as G
from gradus_jl_amd.transfer_functions import device_tracer
ens = G.EnsembleMI355X(0)
kerr = G.KerrMetric(1.0, 0.998)
tab = G.TabulatedMetric(kerr)
x = np.array([0.0, 1000.0, math.radians(40), 0.0])
trs = {"table": device_tracer(tab, x, 2 * x[1], G.chart_for_metric(tab, 2 * x[1]), G.ConstPointFunctions.redshift(tab, x, ensemble=ens), ens),
       "fused": device_tracer(kerr, x, 2 * x[1], G.chart_for_metric(kerr, 2 * x[1]), G.ConstPointFunctions.redshift(kerr, x, ensemble=ens), ens)}
rng = np.random.default_rng(5)
for n in (1, 7, 31, 32, 33, 126, 127, 500):
    rr, th = rng.uniform(2.0, 45.0, n), rng.uniform(0, 2 * math.pi, n)
    al, be = rr * np.cos(th), rr * np.sin(th)
    for name, tr in trs.items():
        first = tr.tangent(al, be)
        bad = 0
        for rep in range(30):
            again = tr.tangent(al, be)
            if not np.array_equal(again, first, equal_nan=True):
                bad += 1
                if bad == 1:
                    d = np.nonzero(~np.isclose(again, first, rtol=0, atol=0, equal_nan=True).all(axis=1))[0]
                    print("   first difference: rays", d[:5], "max abs diff", np.nanmax(np.abs(again - first)))
        # interleave with another launch of a different size in between (other waves' LDS contents change)
        other = tr.tangent(al[: max(1, n // 2)] + 0.1, be[: max(1, n // 2)])
        again = tr.tangent(al, be)
        print(f"n={n:4d} {name}: {bad} of 30 repeats differ; after an unrelated launch identical: {np.array_equal(again, first, equal_nan=True)}; nan rows {int((~np.isfinite(first)).any(axis=1).sum())}")
