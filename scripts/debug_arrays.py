import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G
from oracle import oracle as O
m = G.KerrMetric(1.0, 0.5)
rng = np.random.default_rng(7)
n = 333
xs = np.column_stack([np.zeros(n), rng.uniform(20, 60, n), rng.uniform(0.4, 2.7, n), rng.uniform(0, 6, n)])
vs = np.stack([G.map_impact_parameters(m, x, a, b) for x, a, b in zip(xs, rng.uniform(-8, 8, n), rng.uniform(-8, 8, n))])
cfg = O.make_config("kerr", (1.0, 0.5), disc=(2.0, 30.0), lambda_max=300.0)
ref = O.trace(cfg, xs, vs)
ens = G.EnsembleMI355X(0)
for kernel in (0, 1):
    ens.set("kernel", kernel)
    got = G.tracegeodesics(m, xs, vs, G.ThinDisc(2.0, 30.0), (0.0, 300.0), ensemble=ens)
    mm = got["status"] != ref["status"]
    print("kernel", kernel, "status mismatch", mm.sum(), np.bincount(got["status"],minlength=4), np.bincount(ref["status"],minlength=4))
    ok = ~mm
    rel = np.abs(got["lambda_max"][ok]/ref["lambda_max"][ok]-1)
    bad = np.nonzero(ok)[0][rel>1e-6]
    print(" n bad", len(bad))
    for i in bad[:6]:
        g,r=got[i],ref[i]
        print(i, g["status"], g["lambda_max"], r["lambda_max"], "x0", xs[i], "gx0", g["x_init"], "gx", g["x"], "rx", r["x"])
