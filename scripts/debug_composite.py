import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import gradus_jl_amd as G
from oracle import oracle
import harness as Hh
X_OBS = np.array([0.0, 108.6, math.radians(29.8), 0.0])
params = (1.0, 0.7234, 0.2967)
m = G.KerrNewmanMetric(*params)
a = np.linspace(-30.0, 30.0, 32); b = np.linspace(-30.0, 30.0, 32)
aa, bb = np.meshgrid(a, b)
v = np.stack([G.map_impact_parameters(m, X_OBS, al, be) for al, be in zip(aa.ravel(), bb.ravel())])
for label, d, od in (("thin", G.ThinDisc(3.0, 30.0), (3.0, 30.0)),
                     ("composite", G.ThinDisc(3.0, 30.0) @ G.ThinDisc(35.0, 90.0), {"composite": [(3.0, 30.0), (35.0, 90.0)]})):
    ref = oracle.trace(oracle.make_config("kerr-newman", params, disc=od, lambda_max=300.0), X_OBS, v)
    cfg = G.tracing_configuration(m, X_OBS, v, d, (0.0, 300.0), ensemble=G.EnsembleMI355X.__new__(G.EnsembleMI355X))
    host = Hh.trace_endpoints(G, cfg)
    for kern in (0, 1):
        ens = G.EnsembleMI355X(0, kernel=kern)
        got = G.tracegeodesics(m, X_OBS, v, d, (0.0, 300.0), ensemble=ens)
        for name, arr in (("device", got), ("host", host)):
            mism = arr["status"] != ref["status"]
            ok = ~mism & (ref["status"] != 1)
            rel = np.abs(arr["lambda_max"][ok] / ref["lambda_max"][ok] - 1)
            print(label, kern, name, "status mism", int(mism.sum()), "max rel lambda", rel.max() if rel.size else 0, "n>1e-6", int((rel > 1e-6).sum()),
                  "hist dev", np.bincount(arr["status"], minlength=4), "ref", np.bincount(ref["status"], minlength=4))
