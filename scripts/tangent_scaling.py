#!/usr/bin/env python3
"""gr_ray_tangent_device at several sizes and workgroup shapes: is the launch throughput-bound (time ∝ rays) or
tail-bound (a floor set by its slowest waves)?   python scripts/tangent_scaling.py"""
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import gradus_jl_amd as G
from gradus_jl_amd import _lib
from gradus_jl_amd.rendering import abi_pointfunction
from gradus_jl_amd.tracing import lnr_momentum_to_global_velocity_matrix

ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
cfg = G.tracing_configuration(m, x, np.zeros((1, 4)), G.DatumPlane(0.0), 2000.0, ensemble=ens)
acfg = cfg.abi_config()
apf, keep = abi_pointfunction(G.ConstPointFunctions.redshift(m, x))
dev = torch.device("cuda", 0)
L = _lib.load()
Mx = lnr_momentum_to_global_velocity_matrix(m, cfg.position)
out = {}
for block in (64, 256):
    ens.set("block", 0 if block == 64 else block)
    for S, lims in ((256, None), (512, None), (1024, None), (1024, "offaxis")):
        al = (-60.0, 60.0) if lims is None else (5.0, 60.0)
        aa, bb = np.meshgrid(np.linspace(*al, S), np.linspace(-35.0, 35.0, S))
        d_a = torch.from_numpy(aa.ravel().copy()).to(dev)
        d_b = torch.from_numpy(bb.ravel().copy()).to(dev)
        n = S * S
        rs = _lib.gr_rayset()
        for i in range(4):
            rs.x_obs[i] = float(cfg.position[i])
            for k in range(4):
                rs.Mx[4 * i + k] = float(Mx[i, k])
        rs.alpha, rs.beta, rs.area, rs.n = d_a.data_ptr(), d_b.data_ptr(), None, n
        o = torch.empty(n * 8, dtype=torch.float64, device=dev)
        st = gdev_stats = torch.zeros(11, dtype=torch.int64, device=dev)
        ms = []
        for rep in range(4):
            st.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.gr_ray_tangent_device(ens.ctx.handle, C.byref(acfg), C.byref(rs), C.byref(apf), C.c_void_p(o.data_ptr()),
                                               C.c_void_p(st.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        h = st.cpu().numpy()
        key = f"block{block}_S{S}{'_' + lims if lims else ''}"
        out[key] = {"ms": min(ms[1:]), "rays": n, "ns_per_ray": min(ms[1:]) * 1e6 / n, "steps_per_ray": float(h[1] + h[2]) / max(int(h[0]), 1),
                    "rays_counted": int(h[0])}
        print(key, out[key], flush=True)
print(json.dumps(out))
