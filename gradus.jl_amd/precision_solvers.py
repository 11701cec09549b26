"""Precision solvers on device-traced geodesics -- the public API of src/tracing/precision-solvers.jl.

    find_offset_for_radius            :238-277   offset on the image plane whose ray lands at radius rₑ
    impact_parameters_for_radius[!]   :279-345   a ring of (α, β) for an emission radius
    impact_parameters_for_radius_obscured :347-380  the same, NaN where the disc hides the ring from view
    jacobian_∂αβ_∂gr                  :401-451   |∂(ρ, g)/∂(α, β)|⁻¹
    optimize_for_target / impact_parameters_for_target :453-550  the ray through a given point

MI355X-first: the reference solves each angle of a ring with its own serial Newton iteration on a
re-initialised integrator; here all angles (of all radii) advance in lock-step, one launch of
`gr_ray_summary` per iteration (transfer_functions.find_offsets_for_radius).  Thick discs are traced
against one DatumPlane per emission radius (`gr_rayset.height`).  The target solver replaces the
reference's Nelder-Mead on a noisy closest-approach measure by a Gauss-Newton iteration on the vector
from the target to the ray's point of closest approach, three saved paths (`gr_trace_paths`) per
iteration.
"""
from __future__ import annotations

import math

import numpy as np

from .geometry import DatumPlane, ThinDisc
from .pointfunctions import ConstPointFunctions
from .status import StatusCodes
from .tracing import chart_for_metric, domain_upper_hemisphere, map_impact_parameters, tracegeodesic_paths
from .transfer_functions import device_tracer, find_offsets_for_radius, jacobians


def _is_thick(d):
    return hasattr(d, "cross_section")


def _redshift_pf(m, x, ensemble, redshift_pf):
    if redshift_pf is not None:
        return redshift_pf
    return ConstPointFunctions.redshift(m, x, **({"ensemble": ensemble} if m.metric_id != 0 else {}))


def _datum_tracer(m, x, ensemble, max_time, chart, redshift_pf, solver_opts):
    x = np.asarray(x, dtype=np.float64)
    max_time = 2.0 * x[1] if max_time is None else max_time
    chart = chart_for_metric(m, 2.0 * x[1]) if chart is None else chart
    return device_tracer(m, x, max_time, chart, _redshift_pf(m, x, ensemble, redshift_pf), ensemble, **solver_opts), max_time, chart


def find_offset_for_radius(m, x, d, rₑ, θₒ, *, zero_atol=1e-7, α0=0.0, β0=0.0, max_time=None, chart=None,
                           redshift_pf=None, ensemble=None, return_points=False, **solver_opts):
    """find_offset_for_radius(m, x, d, rₑ, θₒ; zero_atol, α₀, β₀) (precision-solvers.jl:238-277): the
    offset r on the observer's image plane, α = r cos θₒ + α₀, β = r sin θₒ + β₀, whose geodesic meets
    the disc at emission radius rₑ; NaN where there is none.  `rₑ` and `θₒ` broadcast: any number of
    problems is solved in the same launches.  Thin discs are the datum plane z = 0, thick discs the
    plane z = cross_section(d, rₑ) of each problem."""
    scalar = np.ndim(rₑ) == 0 and np.ndim(θₒ) == 0
    rₑ, θₒ = (np.array(a, dtype=np.float64).ravel() for a in np.broadcast_arrays(rₑ, θₒ))
    trace, _, _ = _datum_tracer(m, x, ensemble, max_time, chart, redshift_pf, solver_opts)
    heights = None
    if _is_thick(d):
        heights = np.array([float(d.cross_section(float(r))) for r in rₑ])
    elif isinstance(d, DatumPlane):
        heights = np.full(rₑ.size, float(d.height))
    elif not isinstance(d, ThinDisc):
        raise NotImplementedError(f"find_offset_for_radius: no implementation for {type(d).__name__}")
    r, pts, g = find_offsets_for_radius(trace, rₑ, θₒ, r_min=m.inner_radius(), α0=α0, β0=β0, zero_atol=zero_atol, heights=heights)
    if return_points:
        return (r[0], pts[0], g[0]) if scalar else (r, pts, g)
    return float(r[0]) if scalar else r


def impact_parameters_for_radius(m, x, d, radius, *, N=500, α0=0.0, β0=0.0, **kwargs):
    """impact_parameters_for_radius(m, x, d, radius; N) (:279-345): (α, β) of N angles θ ∈ [0, 2π] whose
    rays land on the ring of radius `radius`; NaN where no offset exists."""
    θ = np.linspace(0.0, 2.0 * math.pi, int(N))
    r = find_offset_for_radius(m, x, d, np.full(θ.size, float(radius)), θ, α0=α0, β0=β0, **kwargs)
    return r * np.cos(θ) + α0, r * np.sin(θ) + β0


def _cartesian_tangent_vector(d, ρ, h=1e-6):
    """_cartesian_tangent_vector (thick-disc.jl:65-72): unit tangent of the surface (ρ, cross_section(ρ))
    in the x-z plane (the reference differentiates with ForwardDiff; central differences here)."""
    dz = (float(d.cross_section(ρ + h)) - float(d.cross_section(ρ - h))) / (2.0 * h)
    v = np.array([1.0, 0.0, dz])
    return v / np.linalg.norm(v)


def _cartesian_surface_normal(d, ρ, ϕ=None):
    """_cartesian_surface_normal (thick-disc.jl:74-81)"""
    t = _cartesian_tangent_vector(d, ρ)
    n = np.array([-t[2], t[1], t[0]])
    if ϕ is None:
        return n
    c, s = math.cos(ϕ), math.sin(ϕ)
    return np.array([c * n[0] - s * n[1], s * n[0] + c * n[1], n[2]])


def impact_parameters_for_radius_obscured(m, x, d, radius, *, N=500, α0=0.0, β0=0.0, max_time=None, chart=None,
                                          redshift_pf=None, ensemble=None, **solver_opts):
    """impact_parameters_for_radius_obscured (:347-380): as above for a thick disc, with NaN for the
    part of the ring the disc itself hides: every ray is traced again against the disc and counts as
    visible when it ends where the datum-plane ray did (`_is_visible`, :382-399: squared Cartesian
    distance below 1e-12)."""
    if not _is_thick(d):
        raise TypeError("impact_parameters_for_radius_obscured needs a thick disc")
    x = np.asarray(x, dtype=np.float64)
    α, β = impact_parameters_for_radius(m, x, d, radius, N=N, α0=α0, β0=β0, max_time=max_time, chart=chart,
                                        redshift_pf=redshift_pf, ensemble=ensemble, **solver_opts)
    ok = np.isfinite(α)
    trace, max_time, chart = _datum_tracer(m, x, ensemble, max_time, chart, redshift_pf, solver_opts)
    thick = device_tracer(m, x, max_time, chart, _redshift_pf(m, x, ensemble, redshift_pf), ensemble, geometry=d, **solver_opts)
    a, b = np.where(ok, α, 0.0), np.where(ok, β, 0.0)
    gp = trace.endpoints(a, b, np.full(a.size, float(d.cross_section(float(radius)))))
    gp_new = thick.endpoints(a, b)

    def cart(p):
        r, θ, ϕ = p["x"][:, 1], p["x"][:, 2], p["x"][:, 3]
        return np.stack([r * np.sin(θ) * np.cos(ϕ), r * np.sin(θ) * np.sin(ϕ), r * np.cos(θ)], axis=1)

    # the re-trace of the reference stops at gp.λ_max: a later hit is no hit
    dist = np.sum((cart(gp) - cart(gp_new)) ** 2, axis=1)
    visible = ok & (dist <= 1e-12)
    return np.where(visible, α, np.nan), np.where(visible, β, np.nan)


def jacobian_αβ_gr(m, x, d, α, β, max_time=None, *, chart=None, redshift_pf=None, ensemble=None, rel_step=3e-4,
                   **solver_opts):
    """jacobian_∂αβ_∂gr(m, x, d, α, β, max_time) (:401-451): |∂(ρ, g)/∂(α, β)|⁻¹ by central differences,
    four rays per point in one launch; thick discs are traced under `domain_upper_hemisphere()` as in
    the reference.  `α`, `β` broadcast."""
    x = np.asarray(x, dtype=np.float64)
    scalar = np.ndim(α) == 0 and np.ndim(β) == 0
    α, β = (np.array(a, dtype=np.float64).ravel() for a in np.broadcast_arrays(α, β))
    max_time = 2.0 * x[1] if max_time is None else max_time
    chart = chart_for_metric(m, 2.0 * x[1]) if chart is None else chart
    geometry = DatumPlane(0.0) if isinstance(d, ThinDisc) else d
    trace = device_tracer(m, x, max_time, chart, _redshift_pf(m, x, ensemble, redshift_pf), ensemble, geometry=geometry,
                          callback=domain_upper_hemisphere() if _is_thick(d) else None, **solver_opts)
    r, θ = np.hypot(α, β), np.arctan2(β, α)
    J = jacobians(trace, r, θ, rel_step=rel_step)
    return float(J[0]) if scalar else J


# ------------------------------------------------------------------------------------------
# the ray through a target point
# ------------------------------------------------------------------------------------------
def _to_cartesian(r, θ, ϕ):
    s = np.sin(θ)
    return np.stack([r * s * np.cos(ϕ), r * s * np.sin(ϕ), r * np.cos(θ)], axis=-1)


def _closest_approach(path, target_cart):
    """Point of a saved path closest to the target (Cartesian distance in the coordinates'
    (r, θ, ϕ), as `_make_target_objective`, :453-500).  Between saved steps the path is the cubic
    Hermite interpolant of positions and velocities; the minimum of the squared distance on the
    bracketing intervals is refined by golden section.
    Returns (vector from target, λ, x(λ), v(λ))."""
    X = _to_cartesian(path.x[:, 1], path.x[:, 2], path.x[:, 3])
    d2 = np.sum((X - target_cart) ** 2, axis=1)
    k = int(np.argmin(d2))

    def hermite(i, s):
        λ0, λ1 = path.λ[i], path.λ[i + 1]
        h = λ1 - λ0
        x0, x1, v0, v1 = path.x[i], path.x[i + 1], path.v[i], path.v[i + 1]
        h00, h10 = 2 * s ** 3 - 3 * s ** 2 + 1, s ** 3 - 2 * s ** 2 + s
        h01, h11 = -2 * s ** 3 + 3 * s ** 2, s ** 3 - s ** 2
        q = h00 * x0 + h10 * h * v0 + h01 * x1 + h11 * h * v1
        dq = ((6 * s ** 2 - 6 * s) * (x0 - x1)) / h + (3 * s ** 2 - 4 * s + 1) * v0 + (3 * s ** 2 - 2 * s) * v1
        return _to_cartesian(q[1], q[2], q[3]), λ0 + s * h, q, dq

    best = (X[k] - target_cart, float(path.λ[k]), path.x[k].copy(), path.v[k].copy(), float(d2[k]))
    gr = 0.5 * (math.sqrt(5.0) - 1.0)
    for i in (k - 1, k):
        if i < 0 or i + 1 >= path.λ.size or path.λ[i + 1] == path.λ[i]:
            continue
        lo, hi = 0.0, 1.0
        f = lambda s: float(np.sum((hermite(i, s)[0] - target_cart) ** 2))
        c, dd = hi - gr * (hi - lo), lo + gr * (hi - lo)
        fc, fd = f(c), f(dd)
        for _ in range(60):
            if fc < fd:
                hi, dd, fd = dd, c, fc
                c = hi - gr * (hi - lo)
                fc = f(c)
            else:
                lo, c, fc = c, dd, fd
                dd = lo + gr * (hi - lo)
                fd = f(dd)
        p, λ, q, dq = hermite(i, 0.5 * (lo + hi))
        v = float(np.sum((p - target_cart) ** 2))
        if v < best[4]:
            best = (p - target_cart, λ, q, dq, v)
    return best[:4]


def optimize_for_target(target, m, x0, *args, p0=None, max_time=None, chart=None, ensemble=None, max_iter=40,
                        d_tol=1e-9, **solver_opts):
    """optimize_for_target(target, m, x0, [d]; ...) (:502-530): impact parameters (α, β) of the geodesic
    from `x0` that passes through `target` = (r, θ, ϕ).  Returns (α, β, GeodesicPoint of that ray,
    accuracy) with `accuracy` the closest-approach distance reached.

    Gauss-Newton on the vector from the target to the ray's point of closest approach, with the
    Jacobian by differences: the ray and two neighbours are traced with every step saved in one launch
    per iteration.  The start is the target's flat-space position on the image plane (the reference
    starts Nelder-Mead from p0 = (0, 0) and keeps whatever it has when the simplex collapses: its
    `accuracy` of 2e-3 ... 5e-3 is that of the optimiser, test/integration/test-precision.jl)."""
    x0 = np.asarray(x0, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    geometry = args[0] if args else None
    max_time = 2.0 * x0[1] if max_time is None else max_time
    T = _to_cartesian(target[0], target[1], target[2])
    if p0 is None:
        # image-plane axes of an observer at (r, θ, ϕ) looking at the origin: α along e_ϕ, β along -e_θ
        θo, ϕo = x0[2], x0[3]
        eϕ = np.array([-math.sin(ϕo), math.cos(ϕo), 0.0])
        eθ = np.array([math.cos(θo) * math.cos(ϕo), math.cos(θo) * math.sin(ϕo), -math.sin(θo)])
        # the reference's α grows towards -e_ϕ (α ≈ -y for an observer on the x axis; cf. the recorded
        # values of test-precision.jl)
        p = np.array([-float(T @ eϕ), -float(T @ eθ)])
    else:
        p = np.array(p0, dtype=np.float64)
    kw = dict(ensemble=ensemble, **solver_opts)
    if chart is not None:
        kw["chart"] = chart
    targs = (geometry, (0.0, max_time)) if geometry is not None else ((0.0, max_time),)

    def residuals(points):
        v = map_impact_parameters(m, x0, points[:, 0], points[:, 1])
        paths = tracegeodesic_paths(m, x0, v, *targs, **kw)
        out = [_closest_approach(pth, T) for pth in paths]
        return np.array([o[0] for o in out]), out, paths

    best = None
    damping = 1.0
    for _ in range(max_iter):
        δ = 1e-4 * max(1.0, float(np.hypot(*p)))
        R, λs, paths = residuals(np.array([p, p + [δ, 0.0], p + [0.0, δ]]))
        dist = float(np.linalg.norm(R[0]))
        if best is None or dist < best[0]:
            best = (dist, p.copy(), paths[0], λs[0])
            damping = min(1.0, damping * 2.0)
        else:
            # overshoot: go back and take a shorter step
            damping *= 0.25
            p = best[1].copy()
            if damping < 1e-6:
                break
            R, λs, paths = residuals(np.array([p, p + [δ, 0.0], p + [0.0, δ]]))
        if best[0] <= d_tol:
            break
        Jm = np.stack([(R[1] - R[0]) / δ, (R[2] - R[0]) / δ], axis=1)      # 3 x 2
        step, *_ = np.linalg.lstsq(Jm, -R[0], rcond=None)
        if not np.all(np.isfinite(step)) or float(np.linalg.norm(step)) < 1e-13 * max(1.0, float(np.hypot(*p))):
            break
        p = p + damping * step
    dist, p, path, (_, λ, xq, vq) = best
    # the GeodesicPoint of the solution: the ray's record with the state at the point of closest approach
    # (the reference terminates the ray there)
    gp = path.point.copy()
    gp["x"], gp["v"], gp["lambda_max"] = xq, vq, λ
    gp["status"] = StatusCodes.NoStatus
    return float(p[0]), float(p[1]), gp, dist


def impact_parameters_for_target(target, m, x0, *args, **kwargs):
    """impact_parameters_for_target (:532-545) -> (α, β, accuracy)"""
    α, β, _, accuracy = optimize_for_target(target, m, x0, *args, **kwargs)
    return α, β, accuracy
