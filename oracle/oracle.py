"""ctypes binding of the CPU ORACLE (oracle/gradus_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (gradus.jl_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRADUS_ORACLE_LIB selects another build of the same source (oracle/Makefile: nofma, asan)
_LIB_PATH = os.environ.get("GRADUS_ORACLE_LIB") or os.path.join(_HERE, "libgradus_oracle.so")

OUT_OF_DOMAIN, WITHIN_INNER_BOUNDARY, INTERSECTED_WITH_GEOMETRY, NO_STATUS = 0, 1, 2, 3
METRIC_KERR, METRIC_JOHANNSEN = 0, 1
METRIC_IDS = {"kerr": 0, "johannsen": 1, "morris-thorne": 2, "bumblebee": 3, "kerr-newman": 4,
              "johannsen-psaltis": 5, "dilaton-axion": 6, "spherical": 7, "kerr-dark-matter": 8,
              "kerr-refractive": 9, "noz": 10,
              "test-bump": 100}      # a stand-in for a user-defined metric (metrics_tmpl.h): Kerr with a smooth bump in g_tt
DISC_NONE, DISC_THIN, DISC_SHAKURA_SUNYAEV, DISC_TABULATED, DISC_TORUS, DISC_DATUM = 0, 1, 2, 3, 4, 5
DISC_ELLIPTICAL, DISC_PRECESSING_THIN = 6, 7
DISC_COMPOSITE = 8
DISC_MESH = 9          # MeshAccretionGeometry (geometry/meshes.jl)
PF_AFFINE_TIME, PF_REDSHIFT, PF_STATUS, PF_R = 0, 1, 2, 3
FILTER_NONE, FILTER_EARLY_TERM, FILTER_INTERSECTED = 0, 1, 2


class DiscComponent(C.Structure):
    _fields_ = [("disc_id", C.c_int32), ("_pad", C.c_int32), ("disc_r_in", C.c_double), ("disc_r_out", C.c_double),
                ("disc_params", C.c_double * 4)]


class Config(C.Structure):
    _fields_ = [
        ("metric_id", C.c_int32),
        ("disc_id", C.c_int32),
        ("params", C.c_double * 8),
        ("r_inner", C.c_double),
        ("r_outer", C.c_double),
        ("disc_r_in", C.c_double),
        ("disc_r_out", C.c_double),
        ("gtol", C.c_double),
        ("lambda0", C.c_double),
        ("lambda1", C.c_double),
        ("abstol", C.c_double),
        ("reltol", C.c_double),
        ("mu", C.c_double),
        ("maxiters", C.c_int64),
        ("upper_hemisphere", C.c_int32),
        ("_pad", C.c_int32),
        ("hemi_delta", C.c_double),
        ("disc_params", C.c_double * 4),
        ("disc_table", C.c_void_p),
        ("disc_table_n", C.c_int64),
        ("chart_table", C.c_void_p),
        ("chart_table_n", C.c_int64),
        ("chart_theta0", C.c_double),
        ("chart_theta1", C.c_double),
        ("q", C.c_double),
        ("count_windings", C.c_int32),
        ("_pad2", C.c_int32),
        ("winding_plane", C.c_double),
        ("comp_n", C.c_int32),           # DISC_COMPOSITE: the geometries of a CompositeGeometry
        ("_pad3", C.c_int32),
        ("comp", DiscComponent * 4),
    ]


POINT_DTYPE = np.dtype(
    [
        ("status", np.int32),
        ("flags", np.int32),
        ("lambda_min", np.float64),
        ("lambda_max", np.float64),
        ("x_init", np.float64, (4,)),
        ("x", np.float64, (4,)),
        ("v_init", np.float64, (4,)),
        ("v", np.float64, (4,)),
    ],
    align=True,
)
assert POINT_DTYPE.itemsize == 152

STATS_DTYPE = np.dtype(
    [("accepted", np.int32), ("rejected", np.int32), ("rhs_evals", np.int32), ("cond_evals", np.int32)]
)


class PF(C.Structure):
    _fields_ = [
        ("pf_id", C.c_int32),
        ("filter_id", C.c_int32),
        ("fill", C.c_double),
        ("r_isco", C.c_double),
        ("n_plunge", C.c_int64),
        ("plunge_r", C.POINTER(C.c_double)),
        ("plunge_vt", C.POINTER(C.c_double)),
        ("plunge_vr", C.POINTER(C.c_double)),
        ("plunge_vphi", C.POINTER(C.c_double)),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("gradus_oracle.c", "gradus_oracle.h", "metrics_tmpl.h")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None
_USER_LIB_PATH = os.path.join(_HERE, "libgradus_oracle_usermetric.so")


def build_usermetric(force: bool = False) -> str:
    """The oracle plus the stand-in for a user-defined metric ("test-bump", oracle/Makefile `usermetric`)."""
    if force or not os.path.exists(_USER_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_USER_LIB_PATH)
        for f in ("gradus_oracle.c", "gradus_oracle.h", "metrics_tmpl.h")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "usermetric"])
    return _USER_LIB_PATH


class user_metric_library:
    """`with oracle.user_metric_library(): ...` -- every call inside goes to libgradus_oracle_usermetric.so, the build that
    knows the metric "test-bump"; outside, the main oracle is untouched (and bit for bit what it was before that metric existed)."""

    def __enter__(self):
        global _lib, _LIB_PATH
        self._saved = (_lib, _LIB_PATH)
        build_usermetric()
        _lib, _LIB_PATH = None, _USER_LIB_PATH
        return lib()

    def __exit__(self, *exc):
        global _lib, _LIB_PATH
        _lib, _LIB_PATH = self._saved
        return False


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        cp = C.POINTER(Config)
        L.orc_metric_jacobian.argtypes = [cp, C.c_double, C.c_double, dp, dp, dp]
        L.orc_geodesic_equation.argtypes = [cp, dp, dp, dp]
        L.orc_constrain_time.argtypes = [cp, dp, dp]
        L.orc_constrain_time.restype = C.c_double
        L.orc_lnrbasis.argtypes = [cp, dp, dp]
        L.orc_lnrframe.argtypes = [cp, dp, dp]
        L.orc_lnr_transform.argtypes = [cp, dp, dp]
        L.orc_map_impact_parameters.argtypes = [cp, dp, C.c_double, C.c_double, dp]
        L.orc_render_velocities.argtypes = [cp, dp] + [C.c_double] * 4 + [C.c_int64] * 4 + [dp]
        L.orc_trace.argtypes = [cp, dp, C.c_int64, dp, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace.restype = C.c_int
        L.orc_isco.argtypes = [cp]
        L.orc_isco.restype = C.c_double
        L.orc_circular_fourvelocity.argtypes = [cp, C.c_double, dp]
        L.orc_apply_pf.argtypes = [cp, C.POINTER(PF), C.c_void_p, C.c_int64, C.c_double, dp, C.c_int]
        L.orc_plunging_table.argtypes = [cp, C.c_double, dp, dp, dp, dp, C.c_int64]
        L.orc_plunging_table.restype = C.c_int64
        L.orc_tsit5_tableau.argtypes = [dp, dp, dp, dp]
        L.orc_max_threads.restype = C.c_int
        L.orc_trace_steps.argtypes = [cp, dp, dp, C.c_void_p, dp, dp, C.c_int64]
        L.orc_trace_steps.restype = C.c_int64
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def inner_radius(M, a, Q=0.0):
    # inner_radius(m::KerrMetric) = M + sqrt(M^2 - a^2), kerr-metric.jl:72 (same for Johannsen,
    # Bumblebee, Johannsen-Psaltis; Kerr-Newman: M + sqrt(M^2 - a^2 - Q^2), kerr-newman-ad.jl:62)
    return M + math.sqrt(M * M - a * a - Q * Q)


def metric_inner_radius(metric, params):
    if metric == "morris-thorne":
        return 0.0                      # morris-thorne-ad.jl:37
    if metric == "kerr-newman":
        return inner_radius(params[0], params[1], params[2])
    if metric == "spherical":
        return 4.0 * np.finfo(np.float64).eps          # minkowski.jl:15
    if metric == "dilaton-axion":       # dilaton-axion-ad.jl:72-75
        M, a, be, b = params[:4]
        bb = 0.0 if be == 0.0 else be / b
        return M + b + math.sqrt((M + b) ** 2 - a * a + be * be - (M - 2 * b) * M * bb * bb)
    return inner_radius(params[0], params[1])


def mesh_table_header(tri):
    """bounding_box(mesh), meshes.jl:24-44: (x_min, x_max, y_min, y_max, z_min, z_max) over every vertex."""
    pts = np.asarray(tri, dtype=np.float64).reshape(-1, 3)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    return np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]])


def jsf_algorithm(V1, V2, V3, Q1, Q2):
    """jsf_algorithm(V₁, V₂, V₃, Q₁, Q₂), intersections.jl:58-101 -> (hit, t)."""
    L = lib()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (V1, V2, V3, Q1, Q2)]
    t = C.c_double(0.0)
    L.orc_jsf.restype = C.c_int
    L.orc_jsf.argtypes = [C.POINTER(C.c_double)] * 5 + [C.POINTER(C.c_double)]
    hit = L.orc_jsf(*[_dp(v) for v in a], C.byref(t))
    return bool(hit), t.value


def make_config(
    metric="kerr",
    params=(1.0, 0.0),
    disc=None,
    lambda_max=2000.0,
    lambda_min=0.0,
    abstol=1e-9,
    reltol=1e-9,
    gtol=1e-2,
    closest_approach=1.01,
    outer_radius=12000.0,
    mu=0.0,
    upper_hemisphere=False,
    hemi_delta=1e-4,
    maxiters=1_000_000,
    q=0.0,
    chart_table=None,
    chart_theta=(0.0, 0.0),
    winding_plane=None,
) -> Config:
    c = Config()
    c.metric_id = METRIC_IDS[metric]
    for i, p in enumerate(params):
        c.params[i] = float(p)
    c.r_inner = metric_inner_radius(metric, params) * closest_approach
    c.r_outer = outer_radius
    if disc is None:
        c.disc_id = DISC_NONE
    elif isinstance(disc, dict) and "composite" in disc:
        # CompositeGeometry(d1, d2, ...): {"composite": [component, ...]}, each component in one of the forms below
        # (thin disc tuple, {"datum": h}, {"ellipse": ...}, Shakura-Sunyaev dict)
        c.disc_id = DISC_COMPOSITE
        comps = disc["composite"]
        c.comp_n = len(comps)
        for k, d in enumerate(comps):
            one = make_config(metric, params, disc=d)
            c.comp[k].disc_id = one.disc_id
            c.comp[k].disc_r_in, c.comp[k].disc_r_out = one.disc_r_in, one.disc_r_out
            for i_par in range(4):          # (not `q`: that is the test particle's charge, an argument of this function)
                c.comp[k].disc_params[i_par] = one.disc_params[i_par]
    elif isinstance(disc, dict) and "mesh" in disc:
        # MeshAccretionGeometry(mesh): {"mesh": triangles (n, 3, 3)}; the constructor's bounding_box (meshes.jl:11-44) in front
        tri = np.ascontiguousarray(disc["mesh"], dtype=np.float64).reshape(-1, 3, 3)
        tab = np.concatenate([mesh_table_header(tri), tri.ravel()])
        c._keep = tab
        c.disc_id = DISC_MESH
        c.disc_table, c.disc_table_n = tab.ctypes.data, tri.shape[0]
    elif isinstance(disc, dict) and "datum" in disc:     # DatumPlane(height)
        c.disc_id = DISC_DATUM
        c.disc_params[0] = float(disc["datum"])
    elif isinstance(disc, dict) and "ellipse" in disc:   # EllipticalDisc: {"ellipse": (inner_radius, semi_major, semi_minor)}
        c.disc_id = DISC_ELLIPTICAL
        c.disc_r_in, c.disc_r_out = float(disc["ellipse"][0]), float("inf")
        c.disc_params[0], c.disc_params[1] = float(disc["ellipse"][1]), float(disc["ellipse"][2])
    elif isinstance(disc, dict) and "precessing" in disc:  # PrecessingDisc(ThinDisc(r_in, r_out), β, γ): {"precessing": (r_in, r_out, β, γ)}
        c.disc_id = DISC_PRECESSING_THIN
        c.disc_r_in, c.disc_r_out = float(disc["precessing"][0]), float(disc["precessing"][1])
        c.disc_params[0], c.disc_params[1] = float(disc["precessing"][2]), float(disc["precessing"][3])
    elif isinstance(disc, dict) and "torus" in disc:     # the reference smoke test's ThickDisc closure
        c.disc_id = DISC_TORUS
        c.disc_r_in, c.disc_r_out = 0.0, float("inf")
        c.disc_params[0], c.disc_params[1] = float(disc["torus"][0]), float(disc["torus"][1])
        c.disc_params[3] = float(disc.get("legacy", 0))     # see disc_condition(): pre-2023 thick-disc semantics
    elif isinstance(disc, dict) and "table" in disc:     # sampled ThickDisc: {"table": heights, "range": (ρ0, ρ1)}
        tab = np.ascontiguousarray(disc["table"], dtype=np.float64)
        c._keep = tab
        c.disc_id = DISC_TABULATED
        c.disc_r_in, c.disc_r_out = 0.0, float("inf")
        c.disc_params[0], c.disc_params[1], c.disc_params[2] = float(disc["range"][0]), float(disc["range"][1]), float(tab.max())
        c.disc_table, c.disc_table_n = tab.ctypes.data, tab.size
        if disc.get("warped"):         # WarpedThinDisc: signed height table over [inner_radius, outer_radius]
            c.disc_r_in, c.disc_r_out = float(disc["range"][0]), float(disc["range"][1])
            c.disc_params[2], c.disc_params[3] = float(np.abs(tab).max()), 1.0
    elif isinstance(disc, dict):       # ShakuraSunyaev: {"mdot": Ṁ/Ṁedd, "inv_eta": 1/η, "inner_radius": r_isco}
        c.disc_id = DISC_SHAKURA_SUNYAEV
        c.disc_r_in, c.disc_r_out = float(disc["inner_radius"]), float("inf")
        c.disc_params[0], c.disc_params[1] = float(disc["mdot"]), float(disc["inv_eta"])
        c.disc_params[3] = float(disc.get("legacy", 0))
    else:
        c.disc_id = DISC_THIN
        c.disc_r_in, c.disc_r_out = float(disc[0]), float(disc[1])
    c.gtol = gtol
    c.lambda0, c.lambda1 = lambda_min, lambda_max
    c.abstol, c.reltol = abstol, reltol
    c.mu = mu
    c.maxiters = maxiters
    c.upper_hemisphere = int(bool(upper_hemisphere))
    c.hemi_delta = hemi_delta
    c.q = q
    if winding_plane is not None:      # TraceWindings(μ, plane_inc)
        c.count_windings, c.winding_plane = 1, float(winding_plane)
    if chart_table is not None:       # PoloidalShapeChart: r_min(θ_k), θ_k uniform on chart_theta
        tab = np.ascontiguousarray(chart_table, dtype=np.float64)
        c._keep_chart = tab
        c.chart_table, c.chart_table_n = tab.ctypes.data, tab.size
        c.chart_theta0, c.chart_theta1 = float(chart_theta[0]), float(chart_theta[1])
    return c


def metric_jacobian(cfg, r, th):
    g, dr, dth = (np.zeros(5) for _ in range(3))
    lib().orc_metric_jacobian(C.byref(cfg), r, th, _dp(g), _dp(dr), _dp(dth))
    return g, dr, dth


def geodesic_equation(cfg, x, v):
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    acc = np.zeros(4)
    lib().orc_geodesic_equation(C.byref(cfg), _dp(x), _dp(v), _dp(acc))
    return acc


def constrain_time(cfg, x, v):
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    return lib().orc_constrain_time(C.byref(cfg), _dp(x), _dp(v))


def lnrbasis(cfg, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros((4, 4))
    lib().orc_lnrbasis(C.byref(cfg), _dp(x), _dp(T))
    return T


def lnrframe(cfg, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros((4, 4))
    lib().orc_lnrframe(C.byref(cfg), _dp(x), _dp(T))
    return T


def lnr_transform(cfg, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros((4, 4))
    lib().orc_lnr_transform(C.byref(cfg), _dp(x), _dp(T))
    return T


def map_impact_parameters(cfg, x, alphas, betas):
    x = np.ascontiguousarray(x, dtype=np.float64)
    alphas = np.atleast_1d(np.asarray(alphas, dtype=np.float64))
    betas = np.atleast_1d(np.asarray(betas, dtype=np.float64))
    out = np.zeros((alphas.size, 4))
    v = np.zeros(4)
    for i, (a, b) in enumerate(zip(alphas, betas)):
        lib().orc_map_impact_parameters(C.byref(cfg), _dp(x), float(a), float(b), _dp(v))
        out[i] = v
    return out


def render_velocities(cfg, x, alims, blims, W, H, i0=0, n=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = W * H - i0 if n is None else n
    v = np.zeros((n, 4))
    lib().orc_render_velocities(
        C.byref(cfg), _dp(x), float(alims[0]), float(alims[1]), float(blims[0]), float(blims[1]), W, H, i0, n, _dp(v)
    )
    return v


def trace(cfg, x, vs, nthreads=0, stats=False):
    """tracegeodesics(m, x, vs, ...; ensemble=EnsembleEndpointThreads()) -> GeodesicPoint array."""
    vs = np.ascontiguousarray(vs, dtype=np.float64).reshape(-1, 4)
    x = np.ascontiguousarray(x, dtype=np.float64)
    N = vs.shape[0]
    stride = 0 if x.ndim == 1 else 4
    out = np.zeros(N, dtype=POINT_DTYPE)
    st = np.zeros(N, dtype=STATS_DTYPE) if stats else None
    rc = lib().orc_trace(
        C.byref(cfg), _dp(x), stride, _dp(vs), N, out.ctypes.data, st.ctypes.data if stats else None, nthreads
    )
    if rc != 0:
        raise RuntimeError(f"orc_trace failed: {rc}")
    return (out, st) if stats else out


def isco(cfg):
    return lib().orc_isco(C.byref(cfg))


def circular_fourvelocity(cfg, r):
    v = np.zeros(4)
    lib().orc_circular_fourvelocity(C.byref(cfg), float(r), _dp(v))
    return v


def plunging_table(cfg, r_isco, cap=100000):
    """PlungingInterpolation (orbit-solving.jl:99-131): returns (r, vt, vr, vphi) sorted by r with
    the smallest-r row dropped (`sortperm(sol[2, :])[2:end]`)."""
    r, vt, vr, vp = (np.zeros(cap) for _ in range(4))
    n = lib().orc_plunging_table(C.byref(cfg), r_isco, _dp(r), _dp(vt), _dp(vr), _dp(vp), cap)
    idx = np.argsort(r[:n], kind="stable")[1:]
    return tuple(np.ascontiguousarray(a[:n][idx]) for a in (r, vt, vr, vp))


def apply_pf(cfg, points, max_time, pf_id=PF_AFFINE_TIME, filter_id=FILTER_NONE, fill=float("nan"), r_isco=0.0,
             plunge=None, nthreads=0):
    pf = PF()
    pf.pf_id, pf.filter_id, pf.fill, pf.r_isco = pf_id, filter_id, fill, r_isco
    keep = None
    if plunge is not None:
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in plunge]
        pf.n_plunge = keep[0].size
        pf.plunge_r, pf.plunge_vt, pf.plunge_vr, pf.plunge_vphi = (_dp(a) for a in keep)
    out = np.zeros(points.shape[0])
    lib().orc_apply_pf(C.byref(cfg), C.byref(pf), points.ctypes.data, points.shape[0], max_time, _dp(out), nthreads)
    return out


def rendergeodesics(cfg, x, alims, blims, W, H, pf_id=PF_AFFINE_TIME, filter_id=FILTER_EARLY_TERM,
                    nthreads=0, return_points=False, **pfkw):
    """rendergeodesics(m, x, [d], λmax; image_width=W, image_height=H, αlims, βlims, pf)
    (rendering.jl:28-54).  Returns the H x W image (image[y, x]; Julia's column-major H×W)."""
    v = render_velocities(cfg, x, alims, blims, W, H)
    pts = trace(cfg, x, v, nthreads=nthreads)
    img = apply_pf(cfg, pts, cfg.lambda1, pf_id=pf_id, filter_id=filter_id, nthreads=nthreads, **pfkw)
    img = img.reshape(W, H).T
    return (img, pts) if return_points else img


def tsit5_tableau():
    c = np.zeros(7)
    a = np.zeros((7, 7))
    bt = np.zeros(7)
    r = np.zeros((7, 4))
    lib().orc_tsit5_tableau(_dp(c), _dp(a), _dp(bt), _dp(r))
    return c, a, bt, r


def trace_steps(cfg, x, v, cap=100000):
    """(point, t[], r[]) of every accepted step of one ray (debugging aid)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(1, dtype=POINT_DTYPE)
    t, r = np.zeros(cap), np.zeros(cap)
    n = lib().orc_trace_steps(C.byref(cfg), _dp(x), _dp(v), out.ctypes.data, _dp(t), _dp(r), cap)
    return out[0], t[:n], r[:n]


def circular_energy(cfg, r):
    """CircularOrbits.energy(m, r) at θ = π/2 (-u_t)."""
    v = circular_fourvelocity(cfg, r)
    g, _, _ = metric_jacobian(cfg, r, math.pi / 2)
    return -(g[0] * v[0] + g[4] * v[3])


def shakura_sunyaev(cfg, eddington_ratio=0.3):
    """ShakuraSunyaev(m; eddington_ratio = 0.3), shakura-sunyaev.jl:35-54."""
    r_isco = isco(cfg)
    eta = 1.0 - circular_energy(cfg, r_isco)
    return {"mdot": eddington_ratio, "inv_eta": 1.0 / eta, "inner_radius": r_isco}


# ---- the oracle on a value + two tangents scalar (oracle/tangent_oracle.cpp): per-ray pin of the tangent kernels ----
_TAN_LIB_PATH = os.path.join(_HERE, "libgradus_oracle_tangent.so")
_tan_lib = None


def build_tangent(force: bool = False) -> str:
    if force or not os.path.exists(_TAN_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_TAN_LIB_PATH)
        for f in ("tangent_oracle.cpp", "gradus_oracle.c", "gradus_oracle.h", "metrics_tmpl.h")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "tangent"])
    return _TAN_LIB_PATH


def ray_tangent(cfg, x_obs, alpha, beta, *, r_isco, max_time, heights=None, norm_with_tangents=False):
    """(g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, t, status) per ray from the oracle integrated on dual numbers -- what
    jacobian_∂αβ_∂gr reads off ForwardDiff.jacobian around tracegeodesics (precision-solvers.jl:401-451).
    `norm_with_tangents`: the step-size controller sees the tangents too (DiffEqBase's norm on Dual state)."""
    global _tan_lib
    if _tan_lib is None:
        _tan_lib = C.CDLL(build_tangent())
        _tan_lib.orct_ray_tangent.restype = C.c_int
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    x_obs = np.ascontiguousarray(x_obs, dtype=np.float64)
    out = np.zeros((alpha.size, 8))
    pf = PF()
    pf.pf_id, pf.filter_id, pf.fill, pf.r_isco, pf.n_plunge = PF_REDSHIFT, FILTER_NONE, float("nan"), float(r_isco), 0
    h = None if heights is None else np.ascontiguousarray(np.broadcast_to(heights, alpha.shape), dtype=np.float64)
    _tan_lib.orct_set_norm(C.c_int(1 if norm_with_tangents else 0))
    rc = _tan_lib.orct_ray_tangent(C.byref(cfg), C.byref(pf), _dp(x_obs), _dp(alpha), _dp(beta),
                                   _dp(h) if h is not None else None, C.c_int64(alpha.size), C.c_double(max_time), _dp(out))
    _tan_lib.orct_set_norm(C.c_int(0))
    if rc != 0:
        raise RuntimeError(f"orct_ray_tangent failed: {rc}")
    return out
