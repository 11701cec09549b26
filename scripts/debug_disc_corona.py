import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
K = G.corona
ens = G.EnsembleMI355X(0)
m, d = G.KerrMetric(1.0, 0.9), G.ThinDisc(0.0, 200.0)
s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
class OneByOne:
    point_source = fixed_position = False
    def __init__(self, model): self.model = model
    def sample_position_velocity(self, mm): return self.model.sample_position_velocity(mm)
mk = lambda: G.DiscCorona(G.SourceVelocities.co_rotating, 6.0, 4.0, seed=17)
n = 4000
prof = {}
for route in ("1", "0"):
    os.environ["GRADUS_MI355X_DEVICE_CORONA"] = route
    for how, model in (("batch", mk()), ("loop", OneByOne(mk()))):
        prof[(route, how)] = K.emissivity_profile(m, d, model, n_samples=n, sampler=s, N=40, ensemble=ens)
ref = prof[("0", "loop")]
for k, p in prof.items():
    ok = np.isfinite(p.ε) & np.isfinite(ref.ε)
    print("device" if k[0] == "1" else "record", k[1], "finite bins", int(np.isfinite(p.ε).sum()), "max rel eps vs record/loop", float(np.max(np.abs(p.ε[ok] / ref.ε[ok] - 1))), "last", p.ε[-3:])
# per-ray: the record route's g with the batch and with the loop
os.environ["GRADUS_MI355X_DEVICE_CORONA"] = "0"
cg_b = K.tracecorona(m, d, mk(), λmax=10000.0, n_samples=n, sampler=s, ensemble=ens)
cg_l = K.tracecorona(m, d, OneByOne(mk()), λmax=10000.0, n_samples=n, sampler=s, ensemble=ens)
print("hits", len(cg_b.geodesic_points), len(cg_l.geodesic_points))
vb, vl = np.asarray(cg_b.source_velocity), np.asarray(cg_l.source_velocity)
print("source velocities max diff", np.abs(vb - vl).max() if vb.shape == vl.shape else (vb.shape, vl.shape))
