#!/usr/bin/env python3
"""The kernels beside the headline one, each as a small loop of launches for rocprofv3 (scripts/profile_pmc.sh with
PROF_CMD="scripts/sibling_workloads.py <which>" PROF_KERNEL=<substring of the kernel name>):

    c4        JohannsenMetric(a=0.7, α13=2, ϵ3=1) 1024², ThinDisc(isco, 50), interpolated redshift   k_trace_lane<JohannsenMetric,1>
    generic   JohannsenPsaltisMetric(a=0.7, ϵ3=1) 1024², ThinDisc, shadow (hand-fused right-hand side)    k_trace_lane<GenericMetricT<5>,1>
    dual      BumblebeeMetric(a=0.25, l=0.5) 1024², ThinDisc, shadow -- a TRUE dual-number metric          k_trace_lane<GenericMetricT<3>,1>
    dual2     MorrisThorneWormhole(b=1) 1024², ThinDisc, shadow (hand-fused since round 4)                k_trace_lane<GenericMetricT<2>,1>
    dual6     DilatonAxion(a=0.5, β=0.3, b=1) 1024², ThinDisc, shadow -- the heaviest dual-number metric k_trace_lane<GenericMetricT<6>,1>
    mesh      Kerr 1024², the bench observer, MeshAccretionGeometry: a ring slab of 3840 triangles            k_trace_lane<KerrFamily<false>,8>
    dual8 / dual9 / dual10   KerrDarkMatter(a=0.5, 2, 20, 10) / KerrRefractive(a=0.5, n=1.1, 20) / NoZMetric(a=0.5, ϵ=0.5), ThinDisc(6, 50), shadow
    tabkerr / tabc4   GR_METRIC_TABULATED: KerrMetric(a=0.998) resp. the C4 Johannsen metric through a table, ThinDisc(isco, 50), redshift   k_trace_lane<TabulatedMetric,1>
    c5        BASELINE config 5 line profile, 4096² polar-plane rays, fp64 tol 1e-9                  k_trace_lane<KerrFamily<false>,1> (tiled rays)
    c5p       the same through the persistent kernel                                                  k_trace_persistent<...>
    c5f32     config 5 with the fp32 kernels at tol 1e-5                                              gr32::k_trace_*
    applypf   apply(pf, cache) on the 2048² end points of the bench plane                             k_apply_pf
    endpoints gr_render_endpoints_device, 2048² Kerr (152-B records)                                  k_trace_lane<KerrFamily<false>,1>
    corona    gr_corona_trace + gr_corona_bin: lamp post h = 10 over ThinDisc(0, 500), 10⁶ sky samples          k_trace_*<KerrFamily<false>,1> + k_corona_*
    tangent   gr_ray_tangent_device, 1024² Kerr rays against the datum plane (value + ∂/∂α + ∂/∂β)     grt::k_trace_lane<KerrFamily<false>,1>

Prints one JSON line {"rays": rays per launch, "launches": n, ...} (bench.log of the profile run); the first 2
launches are warm-up, like bench.py's.
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import gradus_jl_amd as G

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ens = G.EnsembleMI355X(0)
for kv in filter(None, os.environ.get("SIB_KNOBS", "").split(",")):      # e.g. SIB_KNOBS="kernel=0,block=256": launch knobs (gr_ctx_set)
    k_, v_ = kv.split("=")
    ens.set(k_, int(v_))
ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)
ms = []
extra = {}

if which in ("c4", "generic", "dual", "dual6", "dual2", "dual8", "dual9", "dual10", "tabkerr", "tabc4"):
    if which in ("tabkerr", "tabc4"):
        # GR_METRIC_TABULATED: the bench metric / the C4 metric through a piecewise-polynomial table (TAB_GRID = "m_r,n_theta")
        base = G.KerrMetric(1.0, 0.998) if which == "tabkerr" else G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0)
        m_r, n_theta = (int(t) for t in os.environ.get("TAB_GRID", "24,96").split(","))      # (the default grid of TabulatedMetric)
        m = G.TabulatedMetric(base, m_r=m_r, n_theta=n_theta, max_refinements=0)
        x = np.array([0.0, 1000.0, math.radians(75 if which == "tabkerr" else 70), 0.0])
        pf = G.ConstPointFunctions.redshift(m, x, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
        extra = {"grid": [m_r, n_theta], "table_mb": m.table.nbytes / 1e6}
    elif which == "dual2":
        m = G.MorrisThorneWormhole(1.0)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.shadow()
    elif which == "dual":
        m = G.BumblebeeMetric(1.0, 0.25, 0.5)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.shadow()
    elif which == "dual6":
        m = G.DilatonAxion(1.0, 0.5, 0.3, 1.0)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.shadow()
    elif which in ("dual8", "dual9", "dual10"):
        m = {"dual8": G.KerrDarkMatter(1.0, 0.5, 2.0, 20.0, 10.0), "dual9": G.KerrRefractive(1.0, 0.5, 1.1, 20.0),
             "dual10": G.NoZMetric(1.0, 0.5, 0.5)}[which]
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.shadow()
    elif which == "c4":
        m = G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.redshift(m, x, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
    else:
        m = G.JohannsenPsaltisMetric(1.0, 0.7, 1.0)
        x = np.array([0.0, 1000.0, math.radians(70), 0.0])
        pf = G.ConstPointFunctions.shadow()
    S = int(os.environ.get("SIB_SIZE", "1024"))          # (2048: a launch deep enough that its tail does not decide)
    for _ in range(reps):
        disc = (G.ThinDisc(5.0, 50.0) if which == "dual2" else G.ThinDisc(6.0, 50.0) if which in ("dual8", "dual9", "dual10")
                else G.ThinDisc(m.isco(), 50.0))        # (a wormhole has no ISCO; the last three: no isco() on this side)
        _, _, img, st = G.rendergeodesics(m, x, disc, 2000.0, image_width=S, image_height=S,
                                          alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf, ensemble=ens, stats=True)
        ms.append(st["kernel_ms"])
    rays = S * S
elif which == "mesh":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_mesh_geometry import slab

    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    d = G.MeshAccretionGeometry(slab(2.0, 50.0, 10, 96, 1.0))
    pf = G.ConstPointFunctions.affine_time() @ G.ConstPointFunctions.filter_intersected()
    S = int(os.environ.get("SIB_SIZE", "1024"))
    for _ in range(reps):
        _, _, img, st = G.rendergeodesics(m, x, d, 2000.0, image_width=S, image_height=S, alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf,
                                          ensemble=ens, stats=True)
        ms.append(st["kernel_ms"])
    rays = S * S
    extra = {"triangles": len(d), "pixels_hit": int(np.isfinite(img).sum())}
elif which in ("c5", "c5p", "c5f32", "tabc5", "tabc5f32", "tabc5lo"):
    # tabc5 / tabc5f32 / tabc5lo: config 5 on a USER metric (Kerr through the table): fp64 at 1e-9, fp32 kernels at 1e-5, fp64 kernels at 1e-5
    m = G.KerrMetric(1.0, 0.998)
    if which.startswith("tab"):
        m = G.TabulatedMetric(m)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(m.isco(), 250.0)
    N = int(os.environ.get("SIB_SIZE", "4096"))
    plane = G.PolarPlane(G.GeometricGrid(), Nr=N, Nθ=N, r_min=1.0, r_max=250.0)
    bins = np.linspace(0.1, 1.5, 180)
    tol = 1e-9
    if which == "c5p":
        ens.set("kernel", 1)
    if which in ("c5f32", "tabc5f32"):
        ens.set("precision", 32)
        tol = 1e-5
    if which == "tabc5lo":
        tol = 1e-5
    for _ in range(reps):
        xs, ys, st = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0,
                                   ensemble=ens, stats=True, abstol=tol, reltol=tol)
        ms.append(st["kernel_ms"])
    rays = N * N
    extra = {"tol": tol, "steps_per_ray": st["accepted_steps"] / st["rays"]}
elif which in ("applypf", "endpoints"):
    import torch

    from gradus_jl_amd import _lib
    from gradus_jl_amd import device as gdev
    import ctypes as C
    from gradus_jl_amd.rendering import abi_pointfunction

    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
    S = 2048
    cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=S, image_height=S, alpha_lims=ALIMS,
                                 beta_lims=BLIMS, ensemble=ens)
    dev = torch.device("cuda", 0)
    n = S * S
    raw = torch.empty(n * 152, dtype=torch.uint8, device=dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    acfg = cfg.abi_config()
    apf, keep = abi_pointfunction(pf)
    L = _lib.load()
    if which == "endpoints":
        for i in range(reps):
            ev[i][0].record()
            gdev.render_endpoints_device(cfg, raw)
            ev[i][1].record()
    else:
        gdev.render_endpoints_device(cfg, raw)
        for i in range(reps):
            ev[i][0].record()
            _lib.check(L.gr_apply_pointfunction_device(ens.ctx.handle, C.byref(acfg), C.byref(apf), C.c_void_p(raw.data_ptr()), n, 2000.0,
                                                       C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            ev[i][1].record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    rays = n
    extra = {"bytes_per_ray": 160 if which == "applypf" else 152}
elif which in ("tangent", "tabtangent"):
    import ctypes as C

    import torch

    from gradus_jl_amd import _lib
    from gradus_jl_amd.rendering import abi_pointfunction
    from gradus_jl_amd.tracing import lnr_momentum_to_global_velocity_matrix

    m = G.KerrMetric(1.0, 0.998)
    if which == "tabtangent":      # value + two tangents through the table of a user-defined metric (ABI 8)
        m = G.TabulatedMetric(m)
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    S = 1024
    n = S * S
    cfg = G.tracing_configuration(m, x, np.zeros((1, 4)), G.DatumPlane(0.0), 2000.0, ensemble=ens)
    acfg = cfg.abi_config()
    apf, keep = abi_pointfunction(G.ConstPointFunctions.redshift(m, x))
    dev = torch.device("cuda", 0)
    aa, bb = np.meshgrid(np.linspace(*ALIMS, S), np.linspace(*BLIMS, S))
    d_a = torch.from_numpy(aa.ravel().copy()).to(dev)
    d_b = torch.from_numpy(bb.ravel().copy()).to(dev)
    rs = _lib.gr_rayset()
    Mx = lnr_momentum_to_global_velocity_matrix(m, cfg.position)
    for i in range(4):
        rs.x_obs[i] = float(cfg.position[i])
        for k in range(4):
            rs.Mx[4 * i + k] = float(Mx[i, k])
    rs.alpha, rs.beta, rs.area, rs.n = d_a.data_ptr(), d_b.data_ptr(), None, n
    out = torch.empty(n * 8, dtype=torch.float64, device=dev)
    L = _lib.load()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        _lib.check(L.gr_ray_tangent_device(ens.ctx.handle, C.byref(acfg), C.byref(rs), C.byref(apf), C.c_void_p(out.data_ptr()), None,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ev[i][1].record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    rays = n
    o = out.view(n, 8)
    extra = {"bytes_per_ray": 16 + 64, "hit_fraction": float((o[:, 7] == 2).double().mean())}
elif which == "corona":
    # emissivity_profile(m, d, LampPostModel(h = 10); n_samples = 10⁶, golden-spiral EvenSampler on both hemispheres): the sky rays
    # are formed on the device (src_mode 3), traced, reduced (gr_corona_trace) and binned (gr_corona_bin)
    m = G.KerrMetric(1.0, 0.998)
    if os.environ.get("CORONA_TAB"):          # the same metric through a table
        m = G.TabulatedMetric(m)
    d = G.ThinDisc(0.0, 500.0)
    model = G.LampPostModel(h=10.0)
    if os.environ.get("CORONA_MODEL") == "disc":          # a source with a position per sample (28 doubles per sample in)
        model = G.DiscCorona(G.SourceVelocities.co_rotating, 10.0, 5.0, seed=1)
    elif os.environ.get("CORONA_MODEL") == "ring":
        model = G.RingCorona(G.SourceVelocities.co_rotating, 8.0, 6.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    n = int(os.environ.get("CORONA_SAMPLES", "1000000"))
    call = []
    for i in range(reps):
        prof, st = G.corona.device_radial_profile(m, d, model, sampler=s, n_samples=n, N=100, ensemble=ens, stats=True)
        ms.append(st.kernel_ms)
        call.append(st.call_ms)
    rays = n
    extra = {"call_ms_median_after_warmup": float(np.median(call[2:])), "finite_bins": int(np.isfinite(prof.ε).sum()),
             "steps_per_ray": (st.accepted_steps + st.rejected_steps) / n}
else:
    raise SystemExit(f"unknown workload {which}")

print(json.dumps({"workload": which, "rays": rays, "launches": reps, "ms": ms, "ms_median_after_warmup": float(np.median(ms[2:])),
                  "rays_per_s": rays / float(np.median(ms[2:])) * 1e3, **extra}))
