"""Device time of the end-point path (gr_render_endpoints: 152 B/ray AoS records) next to the fused render (8 B/ray)
on the bench workload.   python scripts/endpoints_time.py [size]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gradus_jl_amd as G
from gradus_jl_amd import device as gdev

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ens = G.EnsembleMI355X(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ens.set(k, int(v))
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=N, image_height=N, alpha_lims=(-60, 60),
                             beta_lims=(-35, 35), ensemble=ens)
dev = torch.device("cuda", 0)
img = torch.empty(N * N, dtype=torch.float64, device=dev)
raw = torch.empty(N * N * 152, dtype=torch.uint8, device=dev)
for name, f in (("fused image", lambda: gdev.render_device(cfg, pf, img)), ("end points ", lambda: gdev.render_endpoints_device(cfg, raw))):
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(name, " ".join(f"{t:.2f}" for t in ts), "ms")
