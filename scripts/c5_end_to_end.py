import sys, os, math, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import gradus_jl_amd as G
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998); u = np.array([0.0, 1000.0, math.radians(60), 0.0]); d = G.ThinDisc(m.isco(), 250.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=4096, Nθ=4096, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)
res = {}
for sep in ("1", "0", "1"):
    os.environ["GRADUS_MI355X_SEPARABLE_RAYS"] = sep
    t0 = time.perf_counter()
    xs, ys, st = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0, ensemble=ens, stats=True)
    dt = time.perf_counter() - t0
    res.setdefault(sep, []).append({"wall_s": dt, "kernel_ms": st["kernel_ms"], "ysum": float(ys.sum()), "y90": float(ys[90])})
    print(sep, dt, st["kernel_ms"], ys[90])
print(json.dumps(res))
