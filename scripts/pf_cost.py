import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gradus_jl_amd as G
from gradus_jl_amd import device as gdev
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998); x = np.array([0.0, 1000.0, math.radians(75), 0.0]); d = G.ThinDisc(m.isco(), 50.0)
cfg = G.render_configuration(m, x, d, 2000.0, image_width=2048, image_height=2048, alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens)
CPF = G.ConstPointFunctions
pfs = {"redshift": CPF.redshift(m, x) @ CPF.filter_intersected(), "affine": CPF.affine_time()}
out = torch.empty(2048 * 2048, dtype=torch.float64, device="cuda")
for thr in (4, 16):
    ens.set("refill_threshold", thr)
    for name, pf in pfs.items():
        ts = []
        for i in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gdev.render_device(cfg, pf, out); b.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(a.elapsed_time(b))
        print(f"thr {thr:2d} pf {name:9s} median {np.median(ts):.3f} ms")
