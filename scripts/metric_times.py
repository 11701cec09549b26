import math, sys, numpy as np
sys.path.insert(0, "/root/repo")
import gradus_jl_amd as G
ens = G.EnsembleMI355X(0)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
for name, m in (("kerr", G.KerrMetric(1.0, 0.6)), ("kerr-newman", G.KerrNewmanMetric(1.0, 0.6, 0.6)), ("johannsen-psaltis", G.JohannsenPsaltisMetric(1.0, 0.6, 0.5))):
    ms = []
    for _ in range(4):
        st = G.rendergeodesics(m, x, G.ThinDisc(6.0, 50.0), 2000.0, image_width=1024, image_height=1024, alpha_lims=(-60, 60),
                               beta_lims=(-35, 35), ensemble=ens, stats=True)[3]
        ms.append(st["kernel_ms"])
    print(name, " ".join(f"{t:.2f}" for t in ms))
