#!/usr/bin/env python3
"""When did every wave of one gr_ray_tangent_device launch run?  Needs a library built with -DGR_WAVE_TIMELINE
(scripts/build_variant.sh WORK <name> -DGR_WAVE_TIMELINE; GRADUS_MI355X_LIB=ab/<name>.so): every wave of the one-ray-per-lane
kernel writes (start, end [100 MHz clock], hardware id, steps of its longest ray).  Prints the distribution of wave lifetimes,
the number of waves resident over time and where the launch's tail comes from.
    python scripts/wave_timeline.py [S=1024] [order=grid|lpt|plane]
order = plane: the bench workload (fused image of an S x S plane, the plain fp64 lane kernel) instead of the tangent launch;
WT_KNOBS="lpt=2,lpt_lane=1" sets context knobs first; WT_TAB=1: the bench metric through a table (order = plane)"""
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

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
order = sys.argv[2] if len(sys.argv) > 2 else "grid"
ens = G.EnsembleMI355X(0)
for kv in filter(None, os.environ.get("WT_KNOBS", "").split(",")):
    k_, v_ = kv.split("=")
    ens.set(k_, int(v_))
m = G.KerrMetric(1.0, 0.998)
if os.environ.get("WT_TAB"):          # the same metric through a table (GR_METRIC_TABULATED): order = plane only
    m = G.TabulatedMetric(m, max_refinements=0)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
dev = torch.device("cuda", 0)
L = _lib.load()
L.gr_debug_set_timeline.argtypes = [C.c_void_p]
n = S * S
if order == "plane":
    from gradus_jl_amd import device as gdev

    geom = G.ThinDisc(m.isco(), 50.0)
    if os.environ.get("WT_MESH"):          # the slab of scripts/sibling_workloads.py mesh: WT_MESH = triangles per ring
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_mesh_geometry import slab

        geom = G.MeshAccretionGeometry(slab(2.0, 50.0, 10, int(os.environ["WT_MESH"]), 1.0))
    pcfg = G.render_configuration(m, x, geom, 2000.0, image_width=S, image_height=S, alpha_lims=(-60.0, 60.0),
                                  beta_lims=(-35.0, 35.0), ensemble=ens)
    ppf = (G.ConstPointFunctions.affine_time() if os.environ.get("WT_MESH") else G.ConstPointFunctions.redshift(m, x)) @ G.ConstPointFunctions.filter_intersected()
    img = torch.empty(n, dtype=torch.float64, device=dev)

    def launch():
        gdev.render_device(pcfg, ppf, img)
    reps = 4          # (a learned tile order needs a recording launch and a sorting one first)
else:
    cfg = G.tracing_configuration(m, x, np.zeros((1, 4)), G.DatumPlane(0.0), 2000.0, ensemble=ens)
    acfg = cfg.abi_config()
    apf, keep = abi_pointfunction(G.ConstPointFunctions.redshift(m, x))
    Mx = lnr_momentum_to_global_velocity_matrix(m, cfg.position)
    aa, bb = np.meshgrid(np.linspace(-60.0, 60.0, S), np.linspace(-35.0, 35.0, S))
    a, b = aa.ravel().copy(), bb.ravel().copy()
    if order == "lpt":        # rays closest to the polar axis / the photon ring first
        k = np.argsort(np.minimum(np.abs(a), np.hypot(a - 2.5, b)), kind="stable")
        a, b = a[k], b[k]
    d_a, d_b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    rs = _lib.gr_rayset()
    for i in range(4):
        rs.x_obs[i] = float(cfg.position[i])
        for q in range(4):
            rs.Mx[4 * i + q] = float(Mx[i, q])
    rs.alpha, rs.beta, rs.area, rs.n = d_a.data_ptr(), d_b.data_ptr(), None, n
    o = torch.empty(n * 8, dtype=torch.float64, device=dev)

    def launch():
        _lib.check(L.gr_ray_tangent_device(ens.ctx.handle, C.byref(acfg), C.byref(rs), C.byref(apf), C.c_void_p(o.data_ptr()), None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    reps = 2
waves_max = 2 * n // 64 + 8
tl = torch.zeros(4 * waves_max, dtype=torch.int64, device=dev)
for rep in range(reps):
    tl.zero_()
    L.gr_debug_set_timeline(C.c_void_p(tl.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
L.gr_debug_set_timeline(None)
ms = e0.elapsed_time(e1)
t = tl.cpu().numpy().reshape(-1, 4)
t = t[t[:, 1] != 0]
t0, t1, hw, steps = t[:, 0], t[:, 1], t[:, 2], t[:, 3] & 0xFFFF
copies = (t[:, 3] >> 16) & 0xFFFF          # tabulated metric only: patches copied, ms spent copying, ms in the global-memory evaluation
t_copy, t_glob = ((t[:, 3] >> 32) & 0xFFFF) * 640e-6, ((t[:, 3] >> 48) & 0xFFFF) * 640e-6
base = t0.min()
st, en = (t0 - base) / 1e5, (t1 - base) / 1e5          # ms (100 MHz clock)
life = en - st
print(f"launch {ms:.2f} ms, {t.shape[0]} waves; wave lifetime ms: mean {life.mean():.3f} median {np.median(life):.3f} p90 {np.percentile(life, 90):.3f} "
      f"p99 {np.percentile(life, 99):.3f} max {life.max():.3f}; longest ray of a wave, steps: median {np.median(steps):.0f} p99 {np.percentile(steps, 99):.0f} max {steps.max()}")
print(f"last wave starts at {st.max():.2f} ms, last wave ends at {en.max():.2f} ms; sum of lifetimes / launch = {life.sum() / en.max():.1f} waves resident on average")
edges = np.linspace(0, en.max(), 21)
res = [(int(((st < hi) & (en > lo)).sum())) for lo, hi in zip(edges[:-1], edges[1:])]
print("waves resident per 5 % slice of the launch:", res)
late = np.argsort(en)[-8:]
print("the waves that end last (start, end, steps of longest ray):", [(round(float(st[i]), 2), round(float(en[i]), 2), int(steps[i])) for i in late])
if os.environ.get("WT_TAB"):
    slow = np.argsort(life)[-12:]
    print("slowest waves (lifetime ms, steps of longest ray, patches copied, ms copying, ms evaluating from global memory):",
          [(round(float(life[i]), 2), int(steps[i]), int(copies[i]), round(float(t_copy[i]), 2), round(float(t_glob[i]), 2)) for i in slow])
    print(f"all waves: copies per wave median {np.median(copies):.0f} mean {copies.mean():.0f}; µs per copy median {np.median(t_copy * 1e3 / np.maximum(copies, 1)):.2f}; "
          f"share of wave lifetime: copying {t_copy.sum() / life.sum():.3f}, global evaluation {t_glob.sum() / life.sum():.3f} ({int((t_glob > 0).sum())} waves)")
per_step = life * 1e3 / np.maximum(steps, 1)
print(f"µs per step of a wave's longest ray: median {np.median(per_step):.2f} p10 {np.percentile(per_step, 10):.2f} p90 {np.percentile(per_step, 90):.2f}; "
      f"for the 1 % longest waves {np.median(per_step[np.argsort(life)[-len(life) // 100:]]):.2f}")
# idle time of the hardware wave slots between two waves (HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13])
slot = ((hw >> 32) << 16) | (hw & 0xF) | (hw & 0x30) | (hw & 0xF00) | (hw & 0x1000) | (hw & 0xE000)
order_ = np.lexsort((t0, slot))
s_sorted, st_s, en_s = slot[order_], st[order_], en[order_]
same = s_sorted[1:] == s_sorted[:-1]
gaps = (st_s[1:] - en_s[:-1])[same] * 1e3          # µs
body = en_s[:-1][same] < 0.9 * en.max()
print(f"wave slots seen: {np.unique(slot).size}; gap between two waves on one slot, µs (before the last 10 % of the launch): median {np.median(gaps[body]):.1f} "
      f"p90 {np.percentile(gaps[body], 90):.1f} p99 {np.percentile(gaps[body], 99):.1f} mean {gaps[body].mean():.1f}; slot-time idle in gaps: "
      f"{100.0 * gaps[body].sum() / (np.unique(slot).size * 0.9 * en.max() * 1e3):.1f} %")
print(json.dumps({"launch_ms": ms, "waves": int(t.shape[0]), "lifetime_ms": {"mean": float(life.mean()), "p99": float(np.percentile(life, 99)), "max": float(life.max())},
                  "steps_longest": {"median": float(np.median(steps)), "max": int(steps.max())}, "last_start_ms": float(st.max()), "resident_per_slice": res}))
