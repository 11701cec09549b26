"""Oracle step counts of a lamp-post corona's sky rays (DESIGN_measurements.md §M18, §M19): how well does a quantity known BEFORE
the trace (from a ray's initial x, v) predict how many steps the ray takes -- i.e. can the longest waves be launched first?
  python scripts/corona_lanes.py [n_samples=1000000] [chunk=4096]
CPU only (test infrastructure: the oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
from oracle import oracle as O

O.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = G.corona
a = 0.998
m, model = G.KerrMetric(1.0, a), G.LampPostModel(h=10.0)
s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
xs, vs, vsrc = K.sample_position_direction_velocity(m, model, s, n)
ocfg = O.make_config("kerr", (1.0, a), disc=(0.0, 500.0), lambda_max=5000.0)
out = []
for first in [int(f * n) for f in (0.0, 0.05, 0.25, 0.45, 0.52, 0.58, 0.7, 0.9)] + [n - chunk]:
    sl = slice(first, first + chunk)
    ref, st = O.trace(ocfg, xs[0], vs[sl], stats=True)
    steps = st["accepted"] + st["rejected"]
    out.append((first, vs[sl].copy(), np.asarray(steps), ref["status"].copy()))
    print(first, "mean", steps.mean(), "min", steps.min(), "max", steps.max(), flush=True)
np.savez("/tmp/corona_lanes.npz", x=xs[0], firsts=[o[0] for o in out], v=np.stack([o[1] for o in out]), steps=np.stack([o[2] for o in out]), status=np.stack([o[3] for o in out]))
