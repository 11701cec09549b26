#!/usr/bin/env python3
"""The two recorded sums of test/transfer-functions/test-thick-disc.jl on the device, at the reference's tolerance and tighter."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
for a, angle, r_e, edd, gold, atol in ((0.998, 75, 3.0, 0.3, 14.64279128586961, 1e-4), (0.2, 20, 5.469668466100368, 0.2, 21.581370829241525, 1e-2)):
    m = G.KerrMetric(1.0, a)
    x = np.array([0.0, 10_000.0, math.radians(angle), 0.0])
    d = G.ShakuraSunyaev.for_metric(m, eddington_ratio=edd)
    for tol in (1e-9, 1e-10, 1e-11, 1e-12):
        tf = G.cunningham_transfer_function(m, x, d, r_e, β0=2.0, ensemble=ens, abstol=tol, reltol=tol)
        s = float(np.nansum(tf.f))
        print(f"a={a} θ={angle}° rₑ={r_e:.3f} tol={tol:g}: sum f = {s:.6f}  recorded {gold:.6f} (atol {atol:g})  diff {s - gold:+.2e}  finite {int(np.isfinite(tf.f).sum())}")
