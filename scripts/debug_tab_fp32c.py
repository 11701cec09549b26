"""Why does the table through the fp32 kernels step over the disc a little more often than the fused fp32 kernels? step counts (debug)"""
import math, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
warnings.simplefilter("ignore")
ens = G.EnsembleMI355X(0)
base = G.KerrNewmanMetric(1.0, 0.0718579825321098, 0.05675719829050267)
tab = G.TabulatedMetric(base)
x = np.array([0.0, 148.9, math.radians(56.0), 0.0])
disc = G.ThinDisc(6.4006398782153875, 14.805560212722366)
chart = G.chart_for_metric(tab, 300.0)
for tol in (7.7e-6, 1e-5, 1e-6):
    for name, m, prec in (("tab32", tab, 32), ("fused32", base, 32), ("tab64", tab, 64), ("fused64", base, 64)):
        for kern in (0, 1):
            ens.set("precision", prec).set("kernel", kern)
            _, _, img, st = G.rendergeodesics(m, x, disc, 297.8, pf=G.ConstPointFunctions.affine_time() @ G.ConstPointFunctions.filter_intersected(),
                                              image_width=60, image_height=93, alpha_lims=(-15.7, 15.7), beta_lims=(-15.7, 15.7), abstol=tol, reltol=tol,
                                              chart=chart, ensemble=ens, stats=True)
            print(f"tol {tol:.1e} {name:8s} kernel {kern}: hits {np.isfinite(img).sum():5d} accepted/ray {st['accepted_steps'] / st['rays']:.2f} rejected/ray {st['rejected_steps'] / st['rays']:.2f} status {st['status_count']}")
ens.set("precision", 64).set("kernel", 2)
