"""The reference's own benchmark suite (benchmark/integrator/benchmark-tracing.jl) on the MI355X
path: single geodesic and 128 x 128 many-geodesic cases, with / without disc, saving every step /
end points only.  Wall time of the blocking host call (H2D, kernel, D2H), best of `reps`; the CPU oracle
(all host threads) is timed beside it on the end-point cases.   python scripts/benchmark_tracing.py"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
m = G.KerrMetric(M=1.0, a=0.0)
u = np.array([0.0, 1000.0, math.radians(75.0), 0.0])
v = G.map_impact_parameters(m, u, np.array([4.0]), np.array([0.0]))[0]
d = G.ThinDisc(0.0, 1000.0)
span = (0.0, 2000.0)
ab = np.array([(a, b) for a in np.linspace(-11.0, 11.0, 128) for b in np.linspace(-11.0, 11.0, 128)])
vs = G.map_impact_parameters(m, u, ab[:, 0], ab[:, 1])
us = np.tile(u, (vs.shape[0], 1))


def best(f, reps=7):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    return min(ts), r


rows = [
    ("single-geodesic/no-disc", lambda: G.tracegeodesic_path(m, u, v, span, ensemble=ens)),
    ("single-geodesic/no-disc-no-save", lambda: G.tracegeodesics(m, u, v, span, ensemble=ens)),
    ("single-geodesic/with-disc", lambda: G.tracegeodesic_path(m, u, v, d, span, ensemble=ens)),
    ("single-geodesic/with-disc-no-save", lambda: G.tracegeodesics(m, u, v, d, span, ensemble=ens)),
    ("many-geodesic/no-disc", lambda: G.tracegeodesic_paths(m, us, vs, span, cap=512, ensemble=ens)),
    ("many-geodesic/no-disc-no-save", lambda: G.tracegeodesics(m, us, vs, span, ensemble=ens)),
    ("many-geodesic/with-disc", lambda: G.tracegeodesic_paths(m, us, vs, d, span, cap=512, ensemble=ens)),
    ("many-geodesic/with-disc-no-save", lambda: G.tracegeodesics(m, us, vs, d, span, ensemble=ens)),
]
print(f"{'case':44s} {'MI355X ms':>10s} {'geodesics/s':>12s}")
for name, f in rows:
    t, r = best(f)
    n = 1 if name.startswith("single") else vs.shape[0]
    extra = ""
    if "many" in name and "save" not in name.split("/")[1].replace("no-save", ""):
        pass
    print(f"{name:44s} {t * 1e3:10.3f} {n / t:12.3e}")
try:
    from oracle import oracle as O

    for tag, disc in (("many-geodesic/no-disc-no-save", None), ("many-geodesic/with-disc-no-save", (0.0, 1000.0))):
        cfg = O.make_config("kerr", (1.0, 0.0), disc=disc, lambda_max=2000.0)
        t, _ = best(lambda: O.trace(cfg, us, vs), reps=3)
        print(f"{tag + '  [CPU oracle, ' + str(O.lib().orc_max_threads()) + ' threads]':44s} {t * 1e3:10.3f} {vs.shape[0] / t:12.3e}")
except Exception as e:      # oracle not built
    print("oracle unavailable:", e)
