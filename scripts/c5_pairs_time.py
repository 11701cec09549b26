"""C5 rays through the (g, ρ)-pairs path (no histogram in the kernel): launch shapes compared."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = G.KerrMetric(1.0, 0.998)
u = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(m.isco(), 250.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=N, Nθ=N, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)
ens = G.EnsembleMI355X(0)
for name, k, b in (("persistent/256", 1, 256), ("lane/64", 0, 64), ("lane/256", 0, 256)):
    ens.set("kernel", k).set("block", b)
    ms = []
    for _ in range(4):
        x, y, st = G.lineprofile(bins, lambda r: r ** -3.0, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0,
                                 ensemble=ens, stats=True)
        ms.append(st["kernel_ms"])
    print(name, " ".join(f"{t:.2f}" for t in ms))
