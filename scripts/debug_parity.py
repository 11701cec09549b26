import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G
from oracle import oracle as O
X_FAR = np.array([0.0, 1000.0, math.radians(75), 0.0])
ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
W = H = 64
for disc in (None, (m.isco(), 50.0)):
    args = (G.ThinDisc(*disc), 2000.0) if disc else (2000.0,)
    _, _, cache = G.prerendergeodesics(m, X_FAR, *args, image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    cfg = O.make_config("kerr", (1.0, 0.998), disc=disc, lambda_max=2000.0)
    ref, st = O.trace(cfg, X_FAR, O.render_velocities(cfg, X_FAR, ALIMS, BLIMS, W, H), stats=True)
    print("disc", disc, "status mism", (got["status"] != ref["status"]).sum(), np.bincount(got["status"], minlength=4), np.bincount(ref["status"], minlength=4))
    for code in range(4):
        sel = (got["status"] == code) & (ref["status"] == code)
        if not sel.any(): continue
        for f in ("x", "v"):
            scale = np.maximum(np.abs(ref[f][sel]), 1.0)
            e = np.abs(got[f][sel] - ref[f][sel]) / scale
            print("  status", code, f, "max err per comp", e.max(axis=0))
        print("  status", code, "lambda rel", np.max(np.abs(got["lambda_max"][sel] / ref["lambda_max"][sel] - 1)))
