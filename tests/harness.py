"""ctypes binding of tests/host_harness.cpp: the HIP integrator compiled for the host (g++)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libhost_harness.so")
SRC = [os.path.join(HERE, "host_harness.cpp"), os.path.join(ROOT, "gradus.jl_amd", "csrc", "gr_device.hpp"),
       os.path.join(ROOT, "gradus.jl_amd", "csrc", "gr_mesh_grid.hpp"), os.path.join(ROOT, "include", "gradus_mi355x.h"),
       os.path.join(ROOT, "gradus.jl_amd", "csrc", "gr_tabmetric.hpp")]


def build():
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC[0]])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.hh_step_log.restype = C.c_int64
    return _lib


def render_endpoints(G, config):
    L = G._lib
    cfg, pl = config.abi_config(), config.abi_plane()
    n = pl.width * pl.height
    rg = L.gr_range(0, n, max(n, 1), 1)
    out = np.zeros(n, dtype=L.POINT_DTYPE)
    lib().hh_render_endpoints(C.byref(cfg), C.byref(pl), C.byref(rg), C.c_void_p(out.ctypes.data))
    return out


def render(G, config, pf):
    from gradus_jl_amd.rendering import abi_pointfunction

    L = G._lib
    cfg, pl = config.abi_config(), config.abi_plane()
    n = pl.width * pl.height
    rg = L.gr_range(0, n, max(n, 1), 1)
    s, keep = abi_pointfunction(pf)
    img = np.zeros(n)
    lib().hh_render(C.byref(cfg), C.byref(pl), C.byref(rg), C.byref(s), C.c_void_p(img.ctypes.data))
    return img.reshape(pl.width, pl.height).T


def trace_endpoints(G, config):
    L = G._lib
    cfg = config.abi_config()
    v = np.ascontiguousarray(config.velocity, dtype=np.float64)
    x = np.ascontiguousarray(config.position, dtype=np.float64)
    n = v.shape[0]
    out = np.zeros(n, dtype=L.POINT_DTYPE)
    lib().hh_trace_endpoints(C.byref(cfg), C.c_void_p(x.ctypes.data), C.c_int64(0 if x.ndim == 1 else 4),
                             C.c_void_p(v.ctypes.data), C.c_int64(n), C.c_void_p(out.ctypes.data))
    return out


def step_log(G, config, i, cap=100000):
    L = G._lib
    cfg, pl = config.abi_config(), config.abi_plane()
    out = np.zeros(1, dtype=L.POINT_DTYPE)
    t, h = np.zeros(cap), np.zeros(cap)
    n = lib().hh_step_log(C.byref(cfg), C.byref(pl), C.c_int64(i), C.c_void_p(out.ctypes.data),
                          C.c_void_p(t.ctypes.data), C.c_void_p(h.ctypes.data), C.c_int64(cap))
    return out[0], t[:n], h[:n]


# ---- the TANGENT flavour of the integrator (real = value + ∂/∂α + ∂/∂β), tests/host_harness_tangent.cpp ----
SO_TAN = os.path.join(HERE, "libhost_harness_tangent.so")
SRC_TAN = [os.path.join(HERE, "host_harness_tangent.cpp"), os.path.join(ROOT, "gradus.jl_amd", "csrc", "gr_device.hpp"),
           os.path.join(ROOT, "gradus.jl_amd", "csrc", "gr_tangent.hpp"), os.path.join(ROOT, "include", "gradus_mi355x.h")]
_lib_tan = None


def lib_tangent():
    global _lib_tan
    if _lib_tan is None:
        if not os.path.exists(SO_TAN) or any(os.path.getmtime(s) > os.path.getmtime(SO_TAN) for s in SRC_TAN):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO_TAN, SRC_TAN[0]])
        _lib_tan = C.CDLL(SO_TAN)
    return _lib_tan


def ray_tangent_separable(G, config, pf, r, cos_t, sin_t, tiled):
    """The same for a separable ray set (gr_rayset.sep_*: α = r_i cos θ_j, β = r_i sin θ_j formed by the kernel code)."""
    from gradus_jl_amd.rendering import abi_pointfunction
    from gradus_jl_amd.tracing import lnr_momentum_to_global_velocity_matrix

    L = G._lib
    cfg = config.abi_config()
    s, keep = abi_pointfunction(pf)
    r, cos_t, sin_t = (np.ascontiguousarray(a, dtype=np.float64) for a in (r, cos_t, sin_t))
    rs = L.gr_rayset()
    Mx = lnr_momentum_to_global_velocity_matrix(config.metric, config.position)
    for i in range(4):
        rs.x_obs[i] = float(config.position[i])
        for k in range(4):
            rs.Mx[4 * i + k] = float(Mx[i, k])
    rs.sep_r, rs.sep_cos, rs.sep_sin = r.ctypes.data, cos_t.ctypes.data, sin_t.ctypes.data
    rs.sep_nr, rs.sep_nt, rs.sep_tiled, rs.n = r.size, cos_t.size, int(tiled), r.size * cos_t.size
    out = np.zeros((rs.n, 8))
    rc = lib_tangent().hht_ray_tangent(C.byref(cfg), C.byref(rs), C.byref(s), C.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    return out


def ray_tangent(G, config, pf, α, β, heights=None):
    """(g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, t, status) per ray, from the host build of the tangent kernels."""
    from gradus_jl_amd.rendering import abi_pointfunction
    from gradus_jl_amd.tracing import lnr_momentum_to_global_velocity_matrix

    L = G._lib
    cfg = config.abi_config()
    s, keep = abi_pointfunction(pf)
    α = np.ascontiguousarray(α, dtype=np.float64)
    β = np.ascontiguousarray(β, dtype=np.float64)
    rs = L.gr_rayset()
    Mx = lnr_momentum_to_global_velocity_matrix(config.metric, config.position)
    for i in range(4):
        rs.x_obs[i] = float(config.position[i])
        for k in range(4):
            rs.Mx[4 * i + k] = float(Mx[i, k])
    rs.alpha, rs.beta, rs.area, rs.n = α.ctypes.data, β.ctypes.data, None, α.size
    if heights is not None:
        h = np.ascontiguousarray(np.broadcast_to(heights, α.shape), dtype=np.float64)
        rs.height = h.ctypes.data
    out = np.zeros((α.size, 8))
    rc = lib_tangent().hht_ray_tangent(C.byref(cfg), C.byref(rs), C.byref(s), C.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    return out


def tangent_tracer(G, a, x, max_time):
    """A `(α, β) -> (points, g)` tracer with `.tangent`, like gradus_jl_amd.transfer_functions.device_tracer's, running
    the host build of the tangent kernels (for the transfer-function host logic without a GPU)."""
    from gradus_jl_amd.transfer_functions import SUMMARY_DTYPE

    m = G.KerrMetric(1.0, a)
    cfg = G.tracing_configuration(m, x, np.zeros((1, 4)), G.DatumPlane(0.0), max_time,
                                  chart=G.chart_for_metric(m, 2 * x[1], closest_approach=1.005))
    pf = G.ConstPointFunctions.redshift(m, x)

    def tangent(al, be, heights=None):
        return ray_tangent(G, cfg, pf, al, be, heights)

    def trace(al, be):
        out = tangent(al, be)
        pts = np.zeros(out.shape[0], dtype=SUMMARY_DTYPE)
        pts["status"] = out[:, 7].astype(np.int32)
        pts["x"][:, 0], pts["x"][:, 1], pts["x"][:, 2] = out[:, 6], out[:, 1], math.pi / 2
        return pts, out[:, 0].copy()

    trace.tangent = tangent
    return m, trace
