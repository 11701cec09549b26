import sys, math, numpy as np
sys.path.insert(0,'/root/repo')
import gradus_jl_amd as G
ens=G.EnsembleMI355X(0)
m=G.KerrMetric(1.0,0.998); u=np.array([0.0,1000.0,math.radians(60),0.0]); d=G.ThinDisc(m.isco(),250.0)
plane=G.PolarPlane(G.GeometricGrid(),Nr=1024,Nθ=1024,r_min=1.0,r_max=250.0)
for tol in (1e-3,1e-5):
    pts=G.tracegeodesics(m,u,plane,d,(0.0,2000.0),ensemble=ens,abstol=tol,reltol=tol,callback=G.domain_upper_hemisphere())
    fl=pts["flags"]&0xFFFF
    print("tol",tol,"flagged",int((fl!=0).sum()),"maxiters",int((fl&1!=0).sum()),"dtmin",int((fl&2!=0).sum()),"nan",int((fl&4!=0).sum()))
    bad=np.flatnonzero(fl!=0)[:5]
    for k in bad: print("   ",k, pts["x"][k], pts["v"][k], pts["lambda_max"][k], fl[k])
