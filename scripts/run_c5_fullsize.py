"""BASELINE config C5 at full size: 4096 x 4096 polar-plane rays, Kerr a=0.998, θ=60°, ThinDisc(isco, 250),
domain_upper_hemisphere, ε = r^-3, bins = range(0.1, 1.5, 180).  Prints timing and profile checks."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = G.KerrMetric(1.0, 0.998)
u = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(m.isco(), 250.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=N, Nθ=N, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)
ens = G.EnsembleMI355X(0)
ref = None
for prec, tol in ((64, 1e-9), (64, 1e-7), (64, 1e-5), (64, 1e-3), (32, 1e-6), (32, 1e-5), (32, 1e-4), (32, 1e-3)):
    ens.set("precision", prec)
    t0 = time.perf_counter()
    x, y, st = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0,
                             ensemble=ens, stats=True, abstol=tol, reltol=tol)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = y
    print(f"fp{prec} tol={tol:g} rays={st['rays']} steps/ray={st['accepted_steps']/st['rays']:.1f} rej/ray={st['rejected_steps']/st['rays']:.2f} "
          f"kernel_ms={st['kernel_ms']:.1f} wall_s={dt:.2f} rays/s(kernel)={st['rays']/st['kernel_ms']*1e3:.3e} sum={y.sum():.6f} "
          f"L1 vs 1e-9={np.abs(y-ref).sum():.3e} Linf={np.abs(y-ref).max():.3e} status={st['status_count']} flagged={st['flagged_rays']}")
