"""GR_METRIC_TABULATED through the fp32 kernels against the fused Kerr fp32 kernels and fp64 (debug / measurement)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
ens = G.EnsembleMI355X(0)
X = np.array([0.0, 1000.0, math.radians(75), 0.0])
base = G.KerrMetric(1.0, 0.998)
tm = G.TabulatedMetric(base)
d = G.ThinDisc(base.isco(), 50.0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kw = dict(image_width=size, image_height=size, alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens)
def render(m, prec, tol):
    ens.set("precision", prec)
    pf = G.ConstPointFunctions.redshift(m, X, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
    out = None
    for _ in range(3):
        _, _, img, st = G.rendergeodesics(m, X, d, 2000.0, pf=pf, abstol=tol, reltol=tol, stats=True, **kw)
    ens.set("precision", 64)
    return img, st
def cmp(a, b, la, lb):
    fl = (np.isnan(a) != np.isnan(b)).sum()
    both = ~np.isnan(a) & ~np.isnan(b)
    rel = np.abs(a[both] / b[both] - 1)
    print(f"{la} vs {lb}: flips {fl} ({fl / a.size:.4%}), median {np.median(rel):.2e}, 99% {np.percentile(rel, 99):.2e}, max {rel.max():.2e}")
ref, _ = render(base, 64, 1e-9)
for tol in (1e-5, 1e-6):
    t32, st_t = render(tm, 32, tol)
    f32, st_f = render(base, 32, tol)
    t64, st_t64 = render(tm, 64, tol)
    f64, st_f64 = render(base, 64, tol)
    print(f"tol {tol:g}: kernel ms  tab32 {st_t['kernel_ms']:.2f}  fused32 {st_f['kernel_ms']:.2f}  tab64 {st_t64['kernel_ms']:.2f}  fused64 {st_f64['kernel_ms']:.2f};  flagged tab32 {st_t['flagged_rays']} fused32 {st_f['flagged_rays']}")
    cmp(t32, f32, "tab32", "fused32"); cmp(t32, ref, "tab32", "ref64"); cmp(f32, ref, "fused32", "ref64"); cmp(t64, f64, "tab64", "fused64"); cmp(f64, ref, "fused64@tol", "ref64")
