#!/usr/bin/env python3
"""fp32 kernels against the fp64 kernels on random scenes (all metrics, several geometries, tol 1e-4): status agreement and
end-point agreement at the level single precision allows.  python scripts/soak32.py [n_scenes] [seed]"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradus_jl_amd as G

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
U = lambda a, b: float(rng.uniform(a, b))
ens = G.EnsembleMI355X(0)
fam = [
    (lambda: (1.0, U(-0.998, 0.998)), G.KerrMetric), (lambda: (1.0, U(0, 0.9), U(-1, 2), U(-1, 1), U(-1, 1), U(-1, 2)), G.JohannsenMetric),
    (lambda: (1.0, U(0, 0.19), U(-0.5, 1)), G.BumblebeeMetric), (lambda: (1.0, 0.5, U(0, 0.8)), G.KerrNewmanMetric),
    (lambda: (1.0, U(0, 0.8), U(0, 1)), G.JohannsenPsaltisMetric), (lambda: (U(0.5, 3),), G.MorrisThorneWormhole),
    (lambda: (1.0, U(0.1, 0.8), U(-0.3, 0.3), U(0.3, 1.5)), G.DilatonAxion), (lambda: (), G.SphericalMetric),
    (lambda: (1.0, U(0, 0.9), U(0, 3), U(5, 30), U(5, 20)), G.KerrDarkMatter), (lambda: (1.0, U(0, 0.9), U(0.9, 1.3), U(10, 30)), G.KerrRefractive),
    (lambda: (1.0, U(0, 0.9), U(-0.5, 0.5)), G.NoZMetric),
]
bad, worst, tot, mis, flagged = [], 0.0, 0, 0, 0
conf = np.zeros((4, 4), dtype=np.int64)
for case in range(n_scenes):
    gen, cls = fam[int(rng.integers(0, len(fam)))]
    m = cls(*gen())
    x = np.array([0.0, float(10 ** U(1.3, 3.0)), math.radians(U(10, 170)), 0.0])
    rin = U(1, 8)
    kind = int(rng.integers(0, 4))
    d = [G.ThinDisc(rin, rin + 50), G.ShakuraSunyaev(0.2, 10.0, rin), G.EllipticalDisc(rin, 40.0, 3.0),
         G.PrecessingDisc(G.ThinDisc(rin, rin + 50), 0.3, 1.0)][kind]
    lim = U(8, 50)
    kw = dict(image_width=16, image_height=16, alpha_lims=(-lim, lim), beta_lims=(-lim, lim), abstol=1e-4, reltol=1e-4, ensemble=ens)
    ens.set("kernel", int(rng.integers(0, 2)))
    out = {}
    for prec in (64, 32):
        ens.set("precision", prec)
        _, _, c = G.prerendergeodesics(m, x, d, 2.5 * x[1], **kw)
        out[prec] = c.points.ravel().copy()
    a, b = out[64], out[32]
    ok = (a["status"] == b["status"]) & (a["status"] >= 2) & (b["flags"] == 0)
    np.add.at(conf, (a["status"], b["status"]), 1)
    mis += int((a["status"] != b["status"]).sum()); tot += a.size; flagged += int((b["flags"] & 0xFFFF != 0).sum())
    if ok.any():
        err = float(np.median(np.abs(a["x"][ok][:, 1:3] - b["x"][ok][:, 1:3]) / np.maximum(np.abs(a["x"][ok][:, 1:3]), 1.0)))
        worst = max(worst, err)
        if not (err < 2e-2):
            bad.append((case, cls.__name__, kind, err))
    if not np.all(np.isfinite(b["x"][b["flags"] == 0])):
        bad.append((case, cls.__name__, kind, "non-finite end point without a flag"))
ens.set("precision", 64)
print(f"scenes={n_scenes} rays={tot} status-mismatches={mis} ({mis / tot:.3%}) fp32-flagged={flagged} worst median rel err={worst:.2e} failing={len(bad)}")
print("status fp64 (rows) x fp32 (cols):\n", conf)
for b_ in bad:
    print("  ", b_)
