"""A/B of launch shapes (one-ray-per-lane at several workgroup sizes vs persistent) on the BASELINE
workloads other than the bench image: C2 (1024² Kerr), C4 (Johannsen 1024²), C5 (line profile, 2048²
polar-plane rays) and an endpoint trace.  Prints kernel ms per shape."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
CPF = G.ConstPointFunctions
shapes = [("lane/64", 0, 64), ("lane/128", 0, 128), ("lane/256", 0, 256), ("persistent/256", 1, 256), ("persistent/64", 1, 64)]


def best(fn, reps=4):
    ms = []
    for _ in range(reps):
        ms.append(fn())
    return min(ms)


def render(m, x, d, pf, W):
    def go():
        return G.rendergeodesics(m, x, d, 2000.0, image_width=W, image_height=W, alpha_lims=(-60, 60), beta_lims=(-35, 35),
                                 pf=pf, ensemble=ens, stats=True)[3]["kernel_ms"]
    return go


m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
pf = CPF.redshift(m, x) @ CPF.filter_intersected()
mj = G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0)
xj = np.array([0.0, 1000.0, math.radians(70), 0.0])
pfj = CPF.redshift(mj, xj, ensemble=ens) @ CPF.filter_intersected()
u5 = np.array([0.0, 1000.0, math.radians(60), 0.0])
plane = G.PolarPlane(G.GeometricGrid(), Nr=2048, Nθ=2048, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)


def c5():
    return G.lineprofile(bins, G.PowerLawEmissivity(3), m, u5, G.ThinDisc(m.isco(), 250.0), G.BinningMethod(), plane=plane, ensemble=ens,
                         stats=True)[2]["kernel_ms"]


def endpoints():
    return G.prerendergeodesics(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=1024, image_height=1024,
                                alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens, stats=True)


work = [("C2 Kerr 1024^2 fused", render(m, x, G.ThinDisc(m.isco(), 50.0), pf, 1024)),
        ("C4 Johannsen 1024^2", render(mj, xj, G.ThinDisc(mj.isco(), 50.0), pfj, 1024)),
        ("C5 line profile 2048^2 rays", c5),
        ("Kerr 256^2 fused (shallow)", render(m, x, G.ThinDisc(m.isco(), 50.0), pf, 256))]
for tag, fn in work:
    row = []
    for name, k, b in shapes:
        ens.set("kernel", k).set("block", b)
        row.append(f"{name} {best(fn):.3f}")
    print(f"{tag}: " + " | ".join(row), flush=True)
