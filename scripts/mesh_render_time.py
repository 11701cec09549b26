#!/usr/bin/env python3
"""What a mesh costs: the bench scene's observer, a 1024² image, (a) no geometry, (b) a ring slab of 480 triangles around the
hole (MeshAccretionGeometry: one pass over the triangle list per wave and accepted step inside the bounding box), (c) a thin
disc of the same extent (ContinuousCallback).  python scripts/mesh_render_time.py [n_phi=24]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import gradus_jl_amd as G
from test_mesh_geometry import slab

n_phi = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
pf = G.ConstPointFunctions.affine_time() @ G.ConstPointFunctions.filter_intersected()
kw = dict(image_width=1024, image_height=1024, alpha_lims=(-60.0, 60.0), beta_lims=(-35.0, 35.0), ensemble=ens, stats=True)
mesh = slab(2.0, 50.0, 10, n_phi, 1.0)
for name, d in (("none", None), (f"mesh of {len(mesh)} triangles", G.MeshAccretionGeometry(mesh)), ("thin disc", G.ThinDisc(2.0, 50.0))):
    ms = []
    for _ in range(4):
        args = (m, x, 2000.0) if d is None else (m, x, d, 2000.0)
        _, _, img, st = G.rendergeodesics(*args, pf=pf if d is not None else G.ConstPointFunctions.affine_time(), **kw)
        ms.append(st["kernel_ms"])
    print(f"{name}: kernel {np.median(ms[1:]):.2f} ms, pixels with a value {int(np.isfinite(img).sum())}")
