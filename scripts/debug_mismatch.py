import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G
from oracle import oracle as O
X_FAR = np.array([0.0, 1000.0, math.radians(75), 0.0])
ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
W = H = 128
disc = (m.isco(), 50.0)
_, _, cache = G.prerendergeodesics(m, X_FAR, G.ThinDisc(*disc), 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
got = np.ascontiguousarray(cache.points.T).ravel()
cfg = O.make_config("kerr", (1.0, 0.998), disc=disc, lambda_max=2000.0)
ref, st = O.trace(cfg, X_FAR, O.render_velocities(cfg, X_FAR, ALIMS, BLIMS, W, H), stats=True)
mm = np.nonzero(got["status"] != ref["status"])[0]
print("mismatches", len(mm))
a = np.linspace(*ALIMS, W); b = np.linspace(*BLIMS, H)
for i in mm:
    xi, yi = divmod(i, H)
    g, r = got[i], ref[i]
    print(f"i={i} a={a[xi]:.3f} b={b[yi]:.3f} | gpu st={g['status']} lam={g['lambda_max']:.6f} r={g['x'][1]:.6f} th={g['x'][2]:.6f} rho={g['x'][1]*abs(math.sin(g['x'][2])):.5f} | ref st={r['status']} lam={r['lambda_max']:.6f} r={r['x'][1]:.6f} th={r['x'][2]:.6f} rho={r['x'][1]*abs(math.sin(r['x'][2])):.5f} steps={st[i]['accepted']} rej={st[i]['rejected']}")
# typical error stats
ok = (got["status"] == ref["status"]) & (ref["status"] != 1)
e = np.abs(got["x"][ok] - ref["x"][ok]) / np.maximum(np.abs(ref["x"][ok]), 1)
print("median err", np.median(e), "p99", np.percentile(e, 99), "max", e.max())
pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
img = G.apply(pf, cache)
print("redshift min max", np.nanmin(img), np.nanmax(img), "hits", np.isfinite(img).sum())
