"""Segments / axis terms of GR_METRIC_TABULATED through the fp32 kernels (debug / measurement)."""
import math, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
warnings.simplefilter("ignore")
ens = G.EnsembleMI355X(0)
def cmp(a, b, la, lb):
    fl = (np.isnan(a) != np.isnan(b)).sum()
    both = ~np.isnan(a) & ~np.isnan(b)
    rel = np.abs(a[both] / b[both] - 1)
    print(f"  {la} vs {lb}: flips {fl} ({fl / a.size:.4%}), both {both.sum()}, median {np.median(rel):.2e}, 99% {np.percentile(rel, 99):.2e}, max {rel.max():.2e}")
for which in ("kerr-dark-matter", "dilaton-axion"):
    base = {"kerr-dark-matter": G.KerrDarkMatter(1.0, 0.6, 2.0, 8.0, 7.0), "dilaton-axion": G.DilatonAxion(1.0, 0.35, 0.16, 0.33)}[which]
    tm = G.TabulatedMetric(base)
    x = np.array([0.0, 1000.0, math.radians(20.0 if which == "dilaton-axion" else 70.0), 0.0])
    d = G.ThinDisc(6.0, 60.0)
    chart = G.chart_for_metric(tm, 2000.0)
    out = {}
    for name, m, prec, tol in (("tab32", tm, 32, 1e-5), ("fused32", base, 32, 1e-5), ("tab64lo", tm, 64, 1e-5), ("fused64lo", base, 64, 1e-5), ("fused64", base, 64, 1e-9)):
        ens.set("precision", prec)
        out[name] = G.rendergeodesics(m, x, d, 2000.0, pf=G.ConstPointFunctions.affine_time() @ G.ConstPointFunctions.filter_intersected(),
                                      image_width=128, image_height=128, alpha_lims=(-70, 70), beta_lims=(-70, 70), abstol=tol, reltol=tol,
                                      chart=chart, ensemble=ens)[2]
    ens.set("precision", 64)
    print(which)
    cmp(out["tab32"], out["fused32"], "tab32", "fused32"); cmp(out["tab32"], out["fused64"], "tab32", "fused64"); cmp(out["fused32"], out["fused64"], "fused32", "fused64")
    cmp(out["tab64lo"], out["fused64lo"], "tab64@1e-5", "fused64@1e-5"); cmp(out["fused64lo"], out["fused64"], "fused64@1e-5", "fused64")
