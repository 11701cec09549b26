"""The fp32 half of BASELINE C5 for ONE library build (GRADUS_MI355X_LIB): flagged-ray fraction and L1 against fp64 @ 1e-9 at each
tolerance of the sweep.  Used to tell which change of the fused right-hand side moves the fp32 kernels' rounding noise
(profiles/r3n_fp32_flag_ab.txt).

    GRADUS_MI355X_LIB=abv/x.so python scripts/fp32_flag_ab.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
u = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(m.isco(), 250.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=4096, Nθ=4096, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)


def run(prec, tol):
    ens.set("precision", prec)
    try:
        return G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0, ensemble=ens,
                             stats=True, abstol=tol, reltol=tol)
    finally:
        ens.set("precision", 64)


_, ref, st = run(64, 1e-9)
out = [os.path.basename(os.environ.get("GRADUS_MI355X_LIB", "in-tree")), f"fp64@1e-9 {st['kernel_ms']:.1f} ms"]
for tol in (1e-6, 1e-5, 1e-4, 1e-3):
    _, y, st = run(32, tol)
    out.append(f"fp32@{tol:g}: flagged {st['flagged_rays'] / st['rays']:.5f} L1 {np.abs(y - ref).sum():.3e} rej/ray {st['rejected_steps'] / st['rays']:.2f} {st['kernel_ms']:.1f} ms")
print("  ".join(out))
