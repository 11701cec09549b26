#!/usr/bin/env python3
"""Sizes of the launches behind the default TransferFunctionMethod line profile: how many of its sequential launches carry only
a few straggling root finds?   python scripts/tf_launch_hist.py"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
from gradus_jl_amd import transfer_functions as TF
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(0.0, float("inf"))
bins = np.linspace(0.1, 1.5, 180)
G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, x, d, maxrₑ=50.0, ensemble=ens)
TF.LAUNCH_LOG = []
t0 = time.perf_counter()
G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, x, d, maxrₑ=50.0, ensemble=ens)
wall = time.perf_counter() - t0
log = TF.LAUNCH_LOG; TF.LAUNCH_LOG = None
n = np.array([l[1] for l in log]); k = np.array([l[2] for l in log])
print(f"wall {wall:.3f} s, {n.size} launches, {n.sum()} rays, kernel {k.sum():.0f} ms")
for lo, hi in ((1, 1), (2, 4), (5, 16), (17, 64), (65, 256), (257, 1024), (1025, 10**9)):
    sel = (n >= lo) & (n <= hi)
    print(f"  launches of {lo:5d}..{hi if hi < 10**9 else 'inf':>5} rays: {int(sel.sum()):4d}  kernel ms {k[sel].sum():7.1f}  mean ms {k[sel].mean() if sel.any() else 0:.2f}")
print("first 60 launch sizes:", n[:60].tolist())
