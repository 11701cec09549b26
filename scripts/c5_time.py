"""kernel ms of the C5 line profile (fp64, tol 1e-9) at N x N rays, repeated; used for A/B of builds
(GRADUS_MI355X_LIB selects the library)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = G.KerrMetric(1.0, 0.998)
u = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(m.isco(), 250.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=N, Nθ=N, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)
ens = G.EnsembleMI355X(0)
tol = 1e-9
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    if k == "tol":
        tol = float(v)
    else:
        ens.set(k, int(v))
ms = []
for _ in range(reps):
    x, y, st = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0,
                             ensemble=ens, stats=True, abstol=tol, reltol=tol)
    ms.append(st["kernel_ms"])
print(os.environ.get("GRADUS_MI355X_LIB", "in-tree"), N, " ".join(f"{t:.2f}" for t in ms), "steps/ray", st["accepted_steps"] / st["rays"])
