"""Cunningham transfer functions on device-traced geodesics (second half).

Mirror of src/transfer-functions/cunningham-transfer-functions.jl:1-387: discs that the transfer
function solvers see as a datum plane (ThinDisc -> DatumPlane(0), :1-5) and thick discs (one datum plane
per emission radius, visibility re-trace, thick-surface Jacobians, :253-300), with the precision solvers
of src/tracing/precision-solvers.jl:73-236 (offset for a target radius) and :401-451 (Jacobian).

MI355X-first restructuring.  The reference solves one (rₑ, θ) at a time: a Newton iteration in the
image-plane offset r, each evaluation one geodesic carried on dual numbers, then one more
dual-number geodesic for ∂(ρ, g)/∂(α, β), then a golden-section search in θ for g_min / g_max --
all serial, threaded over rₑ only.  Here EVERY pending (rₑ, θ) problem of EVERY emission radius
advances in lock-step: one safeguarded-Newton iteration for the whole batch is one call of
`gr_ray_summary` (two rays per problem: r and r(1+ε), the derivative by differences), the
Jacobians of the whole batch are one call (four rays per problem, central differences) and the
golden-section searches of all radii and of both extrema share their launches as well.  A table of
150 radii costs the same ~200 launches as one radius.

Parity: the converged offsets solve the same equation ρ(r, θ) = rₑ to the same tolerance
(zero_atol = 1e-7), so they do not depend on the iteration path; derivatives by differences agree
with the reference's forward-mode ones to ~1e-6.  Pinned on the reference's recorded values
(test/smoke-tests/cunningham-transfer-functions.jl:25-39, atol 1e-3 there).  `Optim.GoldenSection`
(third party) is restated from its published algorithm.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np

from .geometry import DatumPlane, ThinDisc
from .pointfunctions import ConstPointFunctions
from .status import StatusCodes
from .tracing import chart_for_metric, tracing_configuration

GOLDEN = 0.5 * (3.0 - math.sqrt(5.0))


@dataclass
class CunninghamTransferData:
    """types.jl:1-12"""

    g_star: np.ndarray
    f: np.ndarray
    t: np.ndarray
    gmin: float
    gmax: float
    rₑ: float


SUMMARY_DTYPE = np.dtype([("status", np.int32), ("x", np.float64, (4,))])


# levels of the bisection tree the reference root finder's bracketing traces per launch (1 = one midpoint per launch, the
# sequential loop; results are identical for every depth: tests/test_newton_ad_reference_branches.py)
BRACKET_DEPTH = int(os.environ.get("GRADUS_MI355X_BRACKET_DEPTH", "4"))
BRACKET_RAYS = 16384        # ... and more levels for small groups, while a launch stays below this many rays

# scripts/lineprofile_tf_time.py sets this to a list: every launch of a device tracer then appends (entry point, rays, kernel ms,
# call ms) so that a product's wall time can be split into launches and host work
LAUNCH_LOG = None


def _logged(name, n):
    """(gr_stats to pass to the call or None, function to call after it)"""
    from . import _lib

    if LAUNCH_LOG is None:
        return None, lambda: None
    st = _lib.gr_stats()
    return st, lambda: LAUNCH_LOG.append((name, int(n), st.kernel_ms, st.call_ms))


def device_tracer(m, x, max_time, chart, redshift_pf, ensemble, geometry=None, callback=None, **solver_opts):
    """(α, β[, heights]) arrays -> (ray summaries, g): every ray against `geometry` (default
    DatumPlane(0); with `heights`, one DatumPlane per ray -- datumplane(d, rₑ), datum-plane.jl:14-17)
    on the device.  One launch (`gr_ray_summary`: impact parameters in, 32 B per ray out -- g, ρ, t,
    status); the summaries are presented with the `status` / `x` fields of end-point records
    (x = (t, ρ, π/2, 0)) that the solvers read.  `trace.endpoints(α, β[, heights])` returns the full
    152-B GeodesicPoint records of the same rays (`gr_rayset_endpoints`)."""
    import ctypes as C

    from . import _lib
    from .rendering import abi_pointfunction

    plane = DatumPlane(0.0) if geometry is None else geometry
    config = tracing_configuration(m, x, np.zeros((1, 4)), plane, max_time, chart=chart, ensemble=ensemble,
                                   callback=callback, **solver_opts)
    cfg = config.abi_config()
    pf, keep_pf = abi_pointfunction(redshift_pf)
    L = _lib.load()
    from .tracing import lnr_momentum_to_global_velocity_matrix

    Mx = lnr_momentum_to_global_velocity_matrix(m, config.position)

    def rayset(α, β, heights):
        α = np.ascontiguousarray(α, dtype=np.float64)
        β = np.ascontiguousarray(β, dtype=np.float64)
        rs = _lib.gr_rayset()
        for i in range(4):
            rs.x_obs[i] = float(config.position[i])
            for k in range(4):
                rs.Mx[4 * i + k] = float(Mx[i, k])
        rs.alpha, rs.beta, rs.area, rs.n = α.ctypes.data, β.ctypes.data, None, α.size
        keep = (α, β)
        if heights is not None:
            if not isinstance(plane, DatumPlane):
                raise ValueError("per-ray heights need a DatumPlane")
            h = np.ascontiguousarray(np.broadcast_to(heights, α.shape), dtype=np.float64)
            rs.height = h.ctypes.data
            keep += (h,)
        return rs, keep

    # an ensemble over several devices: the same launches through the *_multi entry points (contiguous shares of the rays)
    ens_ = config.ensemble
    many = ens_.multi
    if many:
        ctx_arr, ctx_stats = _lib.ctx_array(ens_.contexts)
        n_ctx = len(ens_.contexts)

    def trace(α, β, heights=None):
        rs, keep = rayset(α, β, heights)
        out = np.zeros((rs.n, 4))
        if many:
            _lib.check(L.gr_ray_summary_multi(ctx_arr, n_ctx, C.byref(cfg), C.byref(rs), C.byref(pf), out.ctypes.data, ctx_stats))
        else:
            st, done = _logged("gr_ray_summary", rs.n)
            _lib.check(L.gr_ray_summary(config.ensemble.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), out.ctypes.data,
                                        C.byref(st) if st is not None else None))
            done()
        pts = np.zeros(rs.n, dtype=SUMMARY_DTYPE)
        pts["status"] = out[:, 3].astype(np.int32)
        pts["x"][:, 0], pts["x"][:, 1], pts["x"][:, 2] = out[:, 2], out[:, 1], math.pi / 2
        return pts, out[:, 0].copy()

    def endpoints(α, β, heights=None):
        rs, keep = rayset(α, β, heights)
        pts = np.zeros(rs.n, dtype=_lib.POINT_DTYPE)
        if many:
            _lib.check(L.gr_rayset_endpoints_multi(ctx_arr, n_ctx, C.byref(cfg), C.byref(rs), pts.ctypes.data, ctx_stats))
            return pts
        st, done = _logged("gr_rayset_endpoints", rs.n)
        _lib.check(L.gr_rayset_endpoints(config.ensemble.ctx.handle, C.byref(cfg), C.byref(rs), pts.ctypes.data,
                                         C.byref(st) if st is not None else None))
        done()
        return pts

    def tangent(α, β, heights=None):
        """(g, ρ, ∂g/∂α, ∂g/∂β, ∂ρ/∂α, ∂ρ/∂β, t, status) per ray: dual numbers through the integrator on the device
        (`gr_ray_tangent`), what the reference's ForwardDiff.jacobian around tracegeodesics produces."""
        rs, keep = rayset(α, β, heights)
        out = np.zeros((rs.n, 8))
        # ONE kernel shape for every launch of this tracer: the library's default picks the lane-pair shape for small
        # launches and one lane per ray beyond, and the two agree to rounding only (1e-11) -- a root's Newton iterate would then
        # depend on how many other problems share its launch (or, over several devices, on the share size).  The solvers'
        # launches are latency-bound: the pair shape, always.
        prev = ens_.knobs.get("tangent_pairs", 2)
        ens_.set("tangent_pairs", 1)
        try:
            if many:
                _lib.check(L.gr_ray_tangent_multi(ctx_arr, n_ctx, C.byref(cfg), C.byref(rs), C.byref(pf), out.ctypes.data, ctx_stats))
            else:
                st, done = _logged("gr_ray_tangent", rs.n)
                _lib.check(L.gr_ray_tangent(config.ensemble.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), out.ctypes.data,
                                            C.byref(st) if st is not None else None))
                done()
        finally:
            ens_.set("tangent_pairs", prev)
        return out

    # `pf` holds raw pointers into the plunging table's arrays: they live as long as the tracer does, whoever built the point
    # function (a caller that passes a temporary `ConstPointFunctions.redshift(...)` would otherwise leave them dangling)
    trace._keep = (keep_pf, redshift_pf, config, cfg)
    trace.endpoints = endpoints
    # (every metric has the tangent build of its kernels -- a tabulated one since ABI 8: the tangents ride through the table's own
    # polynomials, as the reference's Duals ride through a user's metric_components)
    trace.tangent = tangent
    # rays are nearly free next to a launch's latency here: the solvers may trace points they might not need
    # (BRACKET_DEPTH, GOLDEN_DEPTH); tracers without this mark (the CPU tests' oracle-driven ones) get one level at a time
    trace.speculate = True
    return trace


def _rho(pts):
    return pts["x"][:, 1] * np.abs(np.sin(pts["x"][:, 2]))


def find_offsets_for_radius(trace, r_target, θ, *, r_min, α0=0.0, β0=0.0, zero_atol=1e-7, max_iter=50, eps=1e-5,
                            heights=None, x0=None, polish=True):
    """_find_offset_for_radius (precision-solvers.jl:135-236) for a batch of (r_target, θ) problems.

    Safeguarded Newton on y(r) = ρ(r, θ) - r_target, which is monotonic in r: the start
    max(20, r_target), the Newton update, the bracket kept from points that fall short of the target
    or into the hole, the biased bisection (2·contra + x)/3 when a step leaves the bracket and the
    acceptance tests (|y| <= zero_atol to converge, 1e-4·r_target to be usable) follow the reference;
    dy/dr comes from a second ray at r(1+ε) instead of a dual number.  Once |y| <= zero_atol the
    iteration is continued while it still gains (Newton is quadratic: one or two more launches), down to
    1e-11 r_target: the redshift of neighbouring samples near g_min / g_max differs by ~1e-9, less than
    what a 1e-7 residual in ρ moves g, and the transfer function there divides by that difference.
    `x0` (optional) replaces the reference's cold start by a guess per problem.  (Tried for the golden-section
    searches -- each point started from the offsets of its neighbour: 20 % fewer launches, but the extremal samples,
    where g - g_min is at the integrator's noise floor, then depend on the path and the recorded statistics move by
    1e-3; not used.)
    Returns (r, end points, g) with r = NaN where no usable offset exists."""
    r_target = np.asarray(r_target, dtype=np.float64)
    θ = np.asarray(θ, dtype=np.float64)
    n = r_target.size
    x = np.maximum(20.0, r_target)
    if x0 is not None:
        x0 = np.asarray(x0, dtype=np.float64)
        x = np.where(np.isfinite(x0) & (x0 > 0.0), x0, x)
    lo = np.zeros(n)                      # contra point: known to fall short (or captured)
    hi = np.full(n, np.inf)
    y = np.full(n, np.inf)
    ybest = np.full(n, np.inf)
    xbest = x.copy()
    best_pts, best_g = None, np.full(n, np.nan)
    n_polish = np.zeros(n, dtype=np.int64)
    active = np.arange(n)
    for _ in range(max_iter + 1):
        if active.size == 0:
            break
        xa = x[active]
        rr = np.concatenate([xa, xa * (1.0 + eps)])
        tt = np.concatenate([θ[active], θ[active]])
        if heights is None:
            pts, g = trace(rr * np.cos(tt) + α0, rr * np.sin(tt) + β0)
        else:       # one datum plane per problem (thick discs: precision-solvers.jl:270-277)
            pts, g = trace(rr * np.cos(tt) + α0, rr * np.sin(tt) + β0, np.concatenate([heights[active], heights[active]]))
        k = active.size
        if best_pts is None:
            best_pts = np.zeros(n, dtype=pts.dtype)
        ρ0, ρ1 = _rho(pts[:k]), _rho(pts[k:])
        hit = pts["status"][:k] == StatusCodes.IntersectedWithGeometry
        ya = ρ0 - r_target[active]
        df = (ρ1 - ρ0) / (xa * eps)
        # keep the best point seen; `polished`: already acceptable and no longer improving
        improved = hit & (np.abs(ya) < ybest[active])
        polished = hit & (ybest[active] <= zero_atol) & (np.abs(ya) >= 0.5 * ybest[active])
        ia = active[improved]
        best_pts[ia] = pts[:k][improved]
        best_g[ia] = g[:k][improved]
        xbest[ia] = xa[improved]
        ybest[ia] = np.abs(ya[improved])
        y[active] = ya
        # ... but not for ever: below ~1e-9 r the residual is the integrator's noise, and a lucky factor two is
        # always available there.  Newton squares the error, so three iterations past zero_atol are at the floor.
        n_polish[active] += (hit & (ybest[active] <= zero_atol)).astype(np.int64)
        conv = polished | (n_polish[active] >= 4) | (hit & (np.abs(ya) <= 1e-11 * np.maximum(r_target[active], 1.0)))
        if not polish:      # the reference's loop: stop at the first iterate with |y| <= zero_atol
            conv = hit & (np.abs(ya) <= zero_atol)
        # bracket bookkeeping
        # a ray that neither hits nor is captured has left the chart: its offset is too large
        # (the reference projects such an end point onto the equator and gets y > 0)
        captured = pts["status"][:k] == StatusCodes.WithinInnerBoundary
        below = captured | (hit & (ya < 0.0))
        lo[active] = np.where(below, np.maximum(lo[active], xa), lo[active])
        hi[active] = np.where(~below, np.minimum(hi[active], xa), hi[active])
        with np.errstate(all="ignore"):
            nx = xa - ya / df
        la, ha = lo[active], hi[active]
        near_hole = ρ0 < (r_min + 1.0)
        bad = ~np.isfinite(nx) | (nx <= la) | (nx >= ha) | (~hit) | (near_hole & below)
        bis = np.where(np.isfinite(ha), (2.0 * la + ha) / 3.0, 2.0 * xa)
        # a bracket that has collapsed below what the map can resolve: accept what we have.  The traced map ρ(r) is
        # only piecewise continuous (a change in the number of integration steps moves ρ by a few 1e-7 at tolerance
        # 1e-9); when the target falls into such a jump no offset reaches zero_atol and bisection would run on to
        # rounding.  1e-10 r in the offset is 1e-10 in ρ, three orders below zero_atol.
        stuck = np.isfinite(ha) & ((ha - la) <= 1e-10 * np.maximum(ha, 1.0))
        nx = np.where(bad, bis, nx)
        x[active] = np.where(conv | stuck, xa, nx)
        active = active[~(conv | stuck)]
    r = xbest.copy()
    ok = (best_pts["status"] == StatusCodes.IntersectedWithGeometry) & (ybest <= 1e-4 * r_target)
    r[~ok] = np.nan
    return r, best_pts, best_g


def find_offsets_for_radius_newton_ad(trace, r_target, θ, *, r_min, α0=0.0, β0=0.0, zero_atol=1e-7, max_iter=50,
                                      heights=None, contrapoint_bias=2.0):
    """_find_offset_for_radius (precision-solvers.jl:135-236) restated step for step for a batch of (r_target, θ)
    problems, with the reference's own ingredients: the derivative dρ/dr of the image-plane map from dual numbers carried
    through the integrator (`trace.tangent`; _make_image_plane_mapper :70-130 does the same with ForwardDiff), the cold
    start max(20, r_target), the Newton update, the contrapoint kept from steps that fall short, the biased bisection
    (2·contra + x)/3 when a step lands below zero or within r_min + 1 of the hole, and -- what matters for parity --
    the loop's exit at the FIRST iterate with |ρ - r_target| <= zero_atol.

    Why this exists beside `find_offsets_for_radius` (safeguarded Newton on a difference quotient, every root polished
    to the integrator's noise floor): it is the reference's iteration, so its roots carry the reference's residuals
    (anything up to zero_atol), and it needs one ray per problem per iteration instead of two.  With Jacobians from dual
    numbers the transfer-function statistics do not depend on which of the two finds the roots (agreement 1e-4 or
    better, tests/test_transfer_functions_tangent_host.py); with difference-quotient Jacobians they do, near the extrema
    of g.  Returns (r, summaries, g, tangent rows) -- the last so that the caller can take the Jacobian of the accepted
    ray without tracing it again."""
    r_target = np.asarray(r_target, dtype=np.float64)
    θ = np.asarray(θ, dtype=np.float64)
    n = r_target.size
    cθ, sθ = np.cos(θ), np.sin(θ)

    def step(idx, r):
        α, β = r * cθ[idx] + α0, r * sθ[idx] + β0
        out = trace.tangent(α, β) if heights is None else trace.tangent(α, β, heights[idx])
        df = out[:, 4] * cθ[idx] + out[:, 5] * sθ[idx]
        return out, df, out[:, 1] - r_target[idx]

    x = np.maximum(20.0, r_target)
    contra = np.zeros(n)
    all_idx = np.arange(n)
    point, df, y = step(all_idx, x)
    Δy = np.zeros(n)
    previous = np.zeros((n, 6))
    failed = np.zeros(n, dtype=bool)
    active = np.abs(y) > zero_atol

    def bracket(jc):
        """Roots.find_zero(i -> step(i)[end], (contra, x), atol = zero_atol): bisection on the bracket, finished when
        |y| <= zero_atol; then the reference re-evaluates `step(x)` -- the last midpoint's values here.

        Bisection is one ray per problem per round and 20-26 rounds long, and a round is a launch as long as one ray: these
        loops were 60 % of the launches of a transfer-function table.  The midpoints a bisection can visit form a binary tree
        that depends on nothing but the bracket, so BRACKET_DEPTH levels of it (2^D - 1 midpoints per problem, formed with the
        very expression the sequential loop uses) are traced in ONE launch and the loop then walks down by the signs it
        finds: the same midpoints, the same exit level, the same numbers -- a third (D = 3) or a quarter of the rounds.
        Every problem leaves at ITS first midpoint with |y| <= zero_atol, like the reference's find_zero on one problem
        (until round 4 a group of problems bisected on until all of them were there): what a problem returns does not
        depend on which other problems share its launches."""
        lo_, hi_ = contra[jc].copy(), x[jc].copy()
        mid, pm, dfm, ym = hi_.copy(), point[jc].copy(), df[jc].copy(), y[jc].copy()
        k = jc.size
        live = np.ones(k, dtype=bool)          # problems still bisecting: each stops at ITS first |y| <= zero_atol, as find_zero does
        it = 0
        while it < 60 and live.any():
            # (small groups go deeper: up to BRACKET_RAYS rays per launch, at most 8 levels)
            kl = int(live.sum())
            D = int(BRACKET_DEPTH) if getattr(trace, "speculate", False) or getattr(trace, "speculate_test", False) else 1
            while D > 1 and D < 8 and kl * ((1 << (D + 1)) - 1) <= BRACKET_RAYS:
                D += 1
            D = max(1, min(D, 60 - it))
            jl = np.flatnonzero(live)
            # level l holds 2^l intervals per problem: node (l, q) = the midpoint of interval q of level l
            los, his, mids = [lo_[jl, None]], [hi_[jl, None]], []
            for l in range(D):
                m_l = 0.5 * (los[l] + his[l])
                mids.append(m_l)
                if l + 1 < D:
                    # children of interval q: (lo, mid) = 2q [y >= 0: the root lies below mid], (mid, hi) = 2q + 1 [y < 0]
                    los.append(np.stack([los[l], m_l], axis=2).reshape(kl, -1))
                    his.append(np.stack([m_l, his[l]], axis=2).reshape(kl, -1))
            flat = np.concatenate([m.ravel() for m in mids])
            idx = np.concatenate([np.repeat(jc[jl], m.shape[1]) for m in mids])
            p_all, df_all, y_all = step(idx, flat)
            off, q = 0, np.zeros(kl, dtype=np.int64)
            going = np.ones(kl, dtype=bool)      # of this launch's problems: not yet at their exit level
            for l in range(D):
                w = mids[l].shape[1]
                sel = (off + np.arange(kl) * w + q)[going]
                g_ = jl[going]
                mid[g_], pm[g_], dfm[g_], ym[g_] = flat[sel], p_all[sel], df_all[sel], y_all[sel]
                neg = ym[g_] < 0
                lo_[g_], hi_[g_] = np.where(neg, mid[g_], lo_[g_]), np.where(neg, hi_[g_], mid[g_])
                qn = q.copy()
                qn[going] = 2 * q[going] + neg.astype(np.int64)
                q = np.where(going, qn, 2 * q)
                off += kl * w
                it += 1
                fin = np.abs(ym[g_]) <= zero_atol
                live[g_[fin]] = False
                going[np.flatnonzero(going)[fin]] = False
                if not going.any() or it >= 60:
                    break
            if it >= 60:
                break
        x[jc], y[jc], df[jc], point[jc] = mid, ym, dfm, pm

    i = 0
    while active.any() and i <= max_iter:
        idx = all_idx[active]
        with np.errstate(all="ignore"):
            next_x = x[idx] - y[idx] / df[idx]
        p2, df2, next_y = step(idx, next_x)
        short = (next_x < 0) | ((next_y < 0) & (y[idx] > 0))
        contra[idx] = np.where(short, np.maximum(contra[idx], next_x), contra[idx])
        bis = short & ((next_x < 0) | (p2[:, 1] < (r_min + 1.0)))
        if bis.any():
            jb = idx[bis]
            nb = (contra[jb] * contrapoint_bias + x[jb]) / (1.0 + contrapoint_bias)
            p3, df3, y3 = step(jb, nb)
            next_x[bis], next_y[bis], df2[bis], p2[bis] = nb, y3, df3, p3
        with np.errstate(all="ignore"):
            # "Converge failed": `point, df, next_y = step(next_x)` has overwritten df by now (precision-solvers.jl:176-193),
            # so the test reads the derivative at next_x -- after the bisection step where one was taken.  The loop breaks
            # with x and y as they were and `point` the NEW one (the pair the reference returns if |y| passes worst_accuracy)
            stop = (next_y < 0) & (y[idx] < 0) & ((-y[idx] / df2) < 0)
            next_Δy = (y[idx] - next_y) / y[idx]
            cycle = ~stop & (y[idx] > 0) & np.any(np.abs(previous[idx] - next_Δy[:, None]) <= zero_atol * 100, axis=1)
        point[idx[stop]] = p2[stop]
        df[idx[stop]] = df2[stop]
        # a cycle is finished off by bracketing (Roots.find_zero on (contra, x)) and the loop is left
        if cycle.any():
            bracket(idx[cycle])
        upd = ~stop & ~cycle
        ju = idx[upd]
        x[ju], y[ju], df[ju], Δy[ju] = next_x[upd], next_y[upd], df2[upd], next_Δy[upd]
        point[ju] = p2[upd]
        previous[ju, i % 6] = next_Δy[upd]
        failed[idx[stop]] = True
        failed[idx[cycle]] = True          # left the loop: not iterated further
        active = (np.abs(y) > zero_atol) & ~failed
        i += 1
    if i >= max_iter:
        # "Exceeded max iter": the problems still iterating are bracketed if they are far off (y > 10)
        late = active & (y > 10.0)
        if late.any():
            bracket(all_idx[late])
    status = point[:, 7].astype(np.int32)
    ok = (x >= 0) & (np.abs(y) <= 1e-4 * r_target) & (status == StatusCodes.IntersectedWithGeometry)
    r = np.where(ok, x, np.nan)
    pts = np.zeros(n, dtype=SUMMARY_DTYPE)
    pts["status"] = status
    pts["x"][:, 0], pts["x"][:, 1], pts["x"][:, 2] = point[:, 6], point[:, 1], math.pi / 2
    return r, pts, point[:, 0].copy(), point


def jacobians(trace, r, θ, *, α0=0.0, β0=0.0, rel_step=3e-4, heights=None):
    """|∂(ρ, g)/∂(α, β)|⁻¹ (jacobian_∂αβ_∂gr, precision-solvers.jl:401-451) for a batch.

    With a tracer that offers `tangent` (the device tracer: `gr_ray_tangent`) the four partial derivatives come from
    dual numbers carried through the integrator, one ray per problem, as in the reference (ForwardDiff.jacobian around
    tracegeodesics).  Otherwise by central differences: four rays per problem in one launch."""
    r, θ = np.asarray(r, dtype=np.float64), np.asarray(θ, dtype=np.float64)
    α, β = r * np.cos(θ) + α0, r * np.sin(θ) + β0
    if getattr(trace, "tangent", None) is not None:
        out = trace.tangent(α, β) if heights is None else trace.tangent(α, β, heights)
        ok = out[:, 7].astype(np.int32) == StatusCodes.IntersectedWithGeometry
        with np.errstate(all="ignore"):
            J = np.abs(1.0 / (out[:, 4] * out[:, 3] - out[:, 5] * out[:, 2]))
        return np.where(ok, J, np.nan)
    δ = rel_step * np.maximum(np.abs(r), 1.0)
    pts, g = trace(np.concatenate([α + δ, α - δ, α, α]), np.concatenate([β, β, β + δ, β - δ]))
    n = r.size
    ρ = _rho(pts)
    ok = np.all((pts["status"] == StatusCodes.IntersectedWithGeometry).reshape(4, n), axis=0)
    dρ_dα = (ρ[:n] - ρ[n:2 * n]) / (2 * δ)
    dρ_dβ = (ρ[2 * n:3 * n] - ρ[3 * n:]) / (2 * δ)
    dg_dα = (g[:n] - g[n:2 * n]) / (2 * δ)
    dg_dβ = (g[2 * n:3 * n] - g[3 * n:]) / (2 * δ)
    with np.errstate(all="ignore"):
        J = np.abs(1.0 / (dρ_dα * dg_dβ - dρ_dβ * dg_dα))
    return np.where(ok, J, np.nan)


class _Workhorse:
    """_rear_workhorse for a datum plane (cunningham-transfer-functions.jl:229-251): (g, J, t) for a
    batch of (rₑ, θ), failing loudly like the reference when an offset cannot be found."""

    def __init__(self, trace, r_min, setup):
        self.trace, self.r_min, self.s = trace, r_min, setup

    def __call__(self, rₑ, θ, x0=None):
        s = self.s
        if s.get("root_finder") == "reference" and getattr(self.trace, "tangent", None) is not None:
            # the reference's iterates: Newton with the dual-number derivative, exit at the first |y| <= zero_atol; the
            # Jacobian of the accepted ray comes with it (same ray, same dual numbers)
            r, pts, g, tan = find_offsets_for_radius_newton_ad(self.trace, rₑ, θ, r_min=self.r_min, α0=s["α0"], β0=s["β0"],
                                                               zero_atol=s["zero_atol"])
            self.last_r = r
            if np.any(np.isnan(r)):
                k = int(np.nonzero(np.isnan(r))[0][0])
                raise RuntimeError(f"Transfer function integration failed (rₑ={np.asarray(rₑ)[k]}, θ={np.asarray(θ)[k]}).")
            with np.errstate(all="ignore"):
                J = np.abs(1.0 / (tan[:, 4] * tan[:, 3] - tan[:, 5] * tan[:, 2]))
            return g, J, pts["x"][:, 0].copy()
        r, pts, g = find_offsets_for_radius(self.trace, rₑ, θ, r_min=self.r_min, α0=s["α0"], β0=s["β0"],
                                            zero_atol=s["zero_atol"], x0=x0, polish=s.get("polish", True))
        self.last_r = r
        if np.any(np.isnan(r)):
            k = int(np.nonzero(np.isnan(r))[0][0])
            raise RuntimeError(f"Transfer function integration failed (rₑ={np.asarray(rₑ)[k]}, θ={np.asarray(θ)[k]}).")
        J = jacobians(self.trace, r, θ, α0=s["α0"], β0=s["β0"])
        return g, J, pts["x"][:, 0].copy()


class _ThickWorkhorse:
    """_rear_workhorse for a thick disc (cunningham-transfer-functions.jl:253-300) for a batch of
    (rₑ, θ): the offset is found against datumplane(d, rₑ) -- one plane per problem, all in the same
    launches --, the ray is re-traced against the disc itself for 1.1 λ_max, and a point whose
    re-trace ends elsewhere (status differs or !isapprox(x, x_thick, rtol = 1e-3)) is obscured: its
    Jacobian is NaN.  Visible points get the Jacobian against the thick surface under
    `domain_upper_hemisphere()` (precision-solvers.jl:401-451)."""

    def __init__(self, datum_trace, thick_trace, jac_trace, d, r_min, setup):
        self.datum, self.thick, self.jac, self.d = datum_trace, thick_trace, jac_trace, d
        self.r_min, self.s = r_min, setup

    def __call__(self, rₑ, θ, x0=None):
        s = self.s
        rₑ, θ = np.asarray(rₑ, dtype=np.float64), np.asarray(θ, dtype=np.float64)
        h = np.array([float(self.d.cross_section(float(r))) for r in rₑ])
        r, _, g = find_offsets_for_radius(self.datum, rₑ, θ, r_min=self.r_min, α0=s["α0"], β0=s["β0"],
                                          zero_atol=s["zero_atol"], heights=h, x0=x0)
        self.last_r = r
        if np.any(np.isnan(r)):
            k = int(np.nonzero(np.isnan(r))[0][0])
            raise RuntimeError(f"Transfer function integration failed (rₑ={rₑ[k]}, θ={θ[k]}).")
        α, β = r * np.cos(θ) + s["α0"], r * np.sin(θ) + s["β0"]
        gp = self.datum.endpoints(α, β, h)
        gp_thick = self.thick.endpoints(α, β)
        # the re-trace runs to 1.1 λ_max only: a later hit would have been left at NoStatus
        same = (gp_thick["status"] == gp["status"]) & (gp_thick["lambda_max"] <= 1.1 * gp["lambda_max"])
        dx = np.linalg.norm(gp["x"] - gp_thick["x"], axis=1)
        close = dx <= 1e-3 * np.maximum(np.linalg.norm(gp["x"], axis=1), np.linalg.norm(gp_thick["x"], axis=1))
        J = jacobians(self.jac, r, θ, α0=s["α0"], β0=s["β0"])
        visible = same & close & np.isfinite(J)
        return g, np.where(visible, J, np.nan), gp["x"][:, 0].copy()


# iterations of the golden-section searches that one round of root finds advances: the points a search can visit next form a
# binary tree that depends on the bracket alone (not on the function values), so GOLDEN_DEPTH levels of it -- 2^D - 1 points per
# search -- are evaluated together and the search then walks down by the comparisons it makes: the same points, the same minima.
GOLDEN_DEPTH = int(os.environ.get("GRADUS_MI355X_GOLDEN_DEPTH", "2"))


def _golden_section_batch(evaluate, record, lower, upper, iterations, depth=None):
    """Optim.optimize(f, lower, upper, GoldenSection(); iterations) for a batch of independent one-dimensional problems.
    `evaluate(idx, x)` evaluates problem idx[k] at x[k] (any number of points per problem in one call) and returns whatever
    `record` needs; `record(x, values)` -- called once per iteration with one point per problem, in problem order -- stores
    the evaluation the search really made and returns f.  Returns the minima found (Optim.minimum)."""
    depth = GOLDEN_DEPTH if depth is None else depth
    lower, upper = np.array(lower, dtype=np.float64), np.array(upper, dtype=np.float64)
    n = lower.size
    every = np.arange(n)
    xmin = lower + GOLDEN * (upper - lower)
    fmin = record(xmin, evaluate(every, xmin))
    it = 0
    while it < iterations:
        D = max(1, min(int(depth), iterations - it))
        # level l: 2^l possible states per problem; child 2q = "the new point was not better", 2q + 1 = "better"
        states, level_pts = [(lower, upper, xmin)], []
        for l in range(D):
            pts, nxt = [], []
            for lo, up, xm in states:
                right = (up - xm) > (xm - lo)
                nx = np.where(right, xm + GOLDEN * (up - xm), xm - GOLDEN * (xm - lo))
                pts.append(nx)
                if l + 1 < D:
                    nxt.append((np.where(~right, nx, lo), np.where(right, nx, up), xm))          # not better
                    nxt.append((np.where(right, xm, lo), np.where(~right, xm, up), nx))          # better
            level_pts.append(pts)
            states = nxt
        flat_x = np.concatenate([p for pts in level_pts for p in pts])
        flat_i = np.tile(every, len(flat_x) // n)
        try:
            vals = evaluate(flat_i, flat_x)
        except RuntimeError:
            # a point the search would not have visited has no solution: go on one level at a time (the on-path point
            # raises as before if it is the one)
            if D == 1:
                raise
            depth = 1
            continue
        off, q = 0, np.zeros(n, dtype=np.int64)
        for l in range(D):
            sel = off + q * n + every                    # problem k's point at node q[k] of this level
            new_x = flat_x[sel]
            new_f = record(new_x, tuple(v[sel] for v in vals))
            right = (upper - xmin) > (xmin - lower)
            better = new_f < fmin
            # right step:  better -> lower = xmin ; else upper = new_x ;  left step mirrored
            lower = np.where(right & better, xmin, np.where(~right & ~better, new_x, lower))
            upper = np.where(right & ~better, new_x, np.where(~right & better, xmin, upper))
            xmin = np.where(better, new_x, xmin)
            fmin = np.where(better, new_f, fmin)
            q = 2 * q + better.astype(np.int64)
            off += n * (1 << l)
            it += 1
    return fmin


def _check_gmin_gmax(gmin, gmax, gs):
    """utils.jl:70-103"""
    if math.isnan(gmin):
        gmin = float(np.min(gs))
    if math.isnan(gmax):
        gmax = float(np.max(gs))
    if gmin == gmax:
        gmin, gmax = float(np.min(gs)), float(np.max(gs))
        if gmin == gmax:
            raise RuntimeError("Cannot use extrema")
    return min(gmin, float(np.min(gs))), max(gmax, float(np.max(gs)))


def cunningham_transfer_functions(m, x, d, radii, *, N=80, N_extrema=17, θ_offset=0.3, zero_atol=1e-7, α0=0.0, β0=0.0,
                                  chart=None, max_time=None, redshift_pf=None, ensemble=None, tracer=None,
                                  thick_tracers=None, polish=True, root_finder="reference", _raw=None, **solver_opts):
    """cunningham_transfer_function (cunningham-transfer-functions.jl:337-387) for every emission
    radius of `radii` at once.  Returns a list of CunninghamTransferData.

    `tracer` replaces the device tracer by another `(α, β) -> (points, g)` callable (the CPU tests
    use it to drive this host logic with oracle-traced rays); the default traces on the MI355X.

    `root_finder = "reference"` (default; needs a tracer with `.tangent`, which the device tracer has): the offsets are
    found by the reference's own Newton iteration with the dual-number derivative and its exit at the first
    |ρ - rₑ| <= zero_atol (`find_offsets_for_radius_newton_ad`), the Jacobian comes from the dual numbers of the
    accepted ray.  `"polished"` (and any tracer without `.tangent`): safeguarded Newton with a difference quotient,
    every root polished to the integrator's noise floor, Jacobians by central differences.  The two agree to 1e-3 in
    the recorded statistics wherever those are well-conditioned; the reference-faithful one reproduces them to
    1e-5 ... 1e-7 there (tests/test_transfer_functions_host.py)."""
    thick = hasattr(d, "cross_section")
    if not thick and not isinstance(d, (ThinDisc, DatumPlane)):
        raise NotImplementedError(f"transfer functions: no implementation for {type(d).__name__}")
    x = np.asarray(x, dtype=np.float64)
    radii = np.atleast_1d(np.asarray(radii, dtype=np.float64))
    max_time = 2.0 * x[1] if max_time is None else max_time
    chart = chart_for_metric(m, 2.0 * x[1]) if chart is None else chart
    if tracer is None:
        if redshift_pf is None:
            redshift_pf = ConstPointFunctions.redshift(m, x, **({"ensemble": ensemble} if m.metric_id != 0 else {}))
        tracer = device_tracer(m, x, max_time, chart, redshift_pf, ensemble, **solver_opts)
    setup = dict(α0=float(α0), β0=float(β0), zero_atol=float(zero_atol), polish=bool(polish), root_finder=root_finder)
    if thick:
        from .tracing import domain_upper_hemisphere

        if thick_tracers is None:
            mk = lambda geometry, callback: device_tracer(m, x, max_time, chart, redshift_pf, ensemble, geometry=geometry,
                                                          callback=callback, **solver_opts)
            thick_tracers = (mk(d, None), mk(d, domain_upper_hemisphere()))
        work = _ThickWorkhorse(tracer, thick_tracers[0], thick_tracers[1], d, m.inner_radius(), setup)
    else:
        work = _Workhorse(tracer, m.inner_radius(), setup)

    R = radii.size
    M = N + 2 * N_extrema
    K = N // 5
    θs0 = np.concatenate([np.linspace(-2 * θ_offset, 2 * θ_offset, K), np.linspace(-math.pi / 2, 3 * math.pi / 2, N - 2 * K),
                          np.linspace(math.pi - 2 * θ_offset, math.pi + 2 * θ_offset, K)])
    data = np.full((R, 4, M), np.nan)           # rows: θ, g, J, t  (utils.jl:1-30)
    g, J, t = work(np.repeat(radii, N), np.tile(θs0, R))
    data[:, 0, :N] = θs0
    data[:, 1, :N] = g.reshape(R, N)
    data[:, 2, :N] = J.reshape(R, N)
    data[:, 3, :N] = t.reshape(R, N)

    # _search_extremal! (:390-438): golden section for g_min about θ = 0 and for g_max about θ = π,
    # every attempt stored; both searches of every radius share their launches
    slot = [N]
    n_iter = (M - N) // 2 - 1
    rr2 = np.concatenate([radii, radii])
    sign = np.concatenate([np.ones(R), -np.ones(R)])

    def evaluate(idx, θ2):
        """(θ after the pole nudge, g, J, t) at θ2[k] for search idx[k] (searches 0..R-1: g_min, R..2R-1: g_max of radius idx - R)"""
        θ2 = np.array(θ2, dtype=np.float64)
        pole = (np.abs(θ2) < 1e-4) | (np.abs(np.abs(θ2) - math.pi) < 1e-4)
        θ2 = np.where(pole, θ2 + 1e-4, θ2)
        g, J, t = work(rr2[idx], θ2)
        return θ2, g, J, t

    def record(_, vals):
        """one stored call per search and iteration (utils.jl:8-30: the accumulator keeps every point the search evaluated)"""
        θ2, g, J, t = vals
        i = slot[0]
        for half, col in ((slice(0, R), i), (slice(R, 2 * R), M - 1 - (i - N))):
            data[:, 0, col], data[:, 1, col], data[:, 2, col], data[:, 3, col] = θ2[half], g[half], J[half], t[half]
        slot[0] += 1
        return sign * g

    lower = np.concatenate([np.full(R, -θ_offset), np.full(R, math.pi - θ_offset)])
    upper = np.concatenate([np.full(R, θ_offset), np.full(R, math.pi + θ_offset)])
    best = _golden_section_batch(evaluate, record, lower, upper, n_iter,
                                 depth=GOLDEN_DEPTH if getattr(tracer, "speculate", False) else 1)
    gmin_c, gmax_c = best[:R], -best[R:]
    if _raw is not None:
        _raw.append((data.copy(), gmin_c.copy(), gmax_c.copy()))

    out = []
    for k in range(R):
        dk = data[k][:, ~np.isnan(data[k, 0])]
        dk = dk[:, np.argsort(dk[0], kind="stable")]
        gmin, gmax = _check_gmin_gmax(float(gmin_c[k]), float(gmax_c[k]), dk[1])
        Js = (gmax - gmin) * dk[2]
        gstar = (dk[1] - gmin) / (gmax - gmin)
        with np.errstate(invalid="ignore"):
            f = (1.0 / (math.pi * radii[k])) * dk[1] * np.sqrt(gstar * (1.0 - gstar)) * Js
        out.append(CunninghamTransferData(gstar, f, dk[3].copy(), gmin, gmax, float(radii[k])))
    return out


def cunningham_transfer_function(m, x, d, rₑ, **kwargs):
    """cunningham_transfer_function(m, x, d, rₑ; N, chart, ...) for one emission radius."""
    return cunningham_transfer_functions(m, x, d, [rₑ], **kwargs)[0]


# ------------------------------------------------------------------------------------------
# branches, radial interpolation and integration into line profiles
# (cunningham-transfer-functions.jl:45-176, transfer-functions-2d.jl:1-84, integration.jl)
# ------------------------------------------------------------------------------------------
def _interp(t, u, x):
    """NaNLinearInterpolator (interpolations.jl:1-30) on arrays.  Inside the knots and without NaN values -- the
    case of every branch evaluation of the integrators -- that is plain linear interpolation (np.interp, ten times
    faster than the general form: the line-profile integration calls this 12 000 times)."""
    x = np.asarray(x, dtype=np.float64)
    if x.size and t[0] <= x.min() and x.max() <= t[-1] and not np.isnan(u).any():
        return np.interp(x, t, u)
    from .corona import _nan_linear_interp

    return _nan_linear_interp(t, u, x)


@dataclass
class TransferBranches:
    """types.jl:107-128: each branch as its own (g✶ knots, values) pair."""

    upper_g: np.ndarray
    upper_f: np.ndarray
    upper_t: np.ndarray
    lower_g: np.ndarray
    lower_f: np.ndarray
    lower_t: np.ndarray
    gmin: float
    gmax: float
    rₑ: float


def splitbranches(ctf: CunninghamTransferData):
    """cunningham-transfer-functions.jl:105-150 -> (lower g✶, f, t, upper g✶, f, t)"""
    g, f, t = ctf.g_star, ctf.f, ctf.t
    imin, imax = int(np.argmin(g)), int(np.argmax(g))
    i1, i2 = (imin, imax) if imax > imin else (imax, imin)
    if i1 == i2:
        raise RuntimeError(f"Resolved same min/max for rₑ = {ctf.rₑ}")
    b1 = np.arange(i1, i2 + 1)
    b2 = np.concatenate([np.arange(0, i1 + 1), np.arange(i2, g.size)])
    if f[b1][1] > f[b1][0]:
        lo, up = b2, b1
    else:
        lo, up = b1, b2
    return g[lo].copy(), f[lo].copy(), t[lo].copy(), g[up].copy(), f[up].copy(), t[up].copy()


def _sorted_with_adjustments(g1, f1, t1, g2, f2, t2, h):
    """_make_sorted_with_adjustments! (:61-103)"""
    out = []
    I1, I2 = np.argsort(g1, kind="stable"), np.argsort(g2, kind="stable")
    g1, f1, t1, g2, f2, t2 = g1[I1], f1[I1], t1[I1], g2[I2], f2[I2], t2[I2]
    t_lo, t_hi = 0.5 * (t1[0] + t2[0]), 0.5 * (t1[-1] + t2[-1])
    for g, f, t in ((g1, f1, t1), (g2, f2, t2)):
        J = (g < 1.0 - h) & (g > h)
        g, f, t = g[J].copy(), f[J].copy(), t[J].copy()
        t[0], t[-1] = t_lo, t_hi
        g[0], g[-1] = 0.0, 1.0
        out += [g, f, t]
    return out


def interpolate_branches(ctf: CunninghamTransferData, h=1e-6) -> TransferBranches:
    """cunningham-transfer-functions.jl:152-176"""
    lg, lf, lt, ug, uf, ut = _sorted_with_adjustments(*splitbranches(ctf), h)
    return TransferBranches(ug, uf, ut, lg, lf, lt, ctf.gmin, ctf.gmax, ctf.rₑ)


@dataclass
class InterpolatingTransferBranches:
    """transfer-functions-2d.jl:1-84: branches sorted by radius; calling it interpolates in rₑ."""

    branches: list
    radii: np.ndarray
    gmin: np.ndarray
    gmax: np.ndarray

    @staticmethod
    def from_branches(branches):
        branches = sorted(branches, key=lambda b: b.rₑ)
        return InterpolatingTransferBranches(branches, np.array([b.rₑ for b in branches]),
                                             np.array([b.gmin for b in branches]), np.array([b.gmax for b in branches]))

    def inner_radius(self):
        return float(self.radii[0])

    def outer_radius(self):
        return float(self.radii[-1])

    def at(self, r):
        """(gmin, gmax, summed-branch evaluator g✶ -> f_lower + f_upper) at radius r."""
        idx = int(np.clip(np.searchsorted(self.radii, r, side="right") - 1, 0, self.radii.size - 2))
        r1, r2 = self.radii[idx], self.radii[idx + 1]
        w = (r - r1) / (r2 - r1)
        b1, b2 = self.branches[idx], self.branches[idx + 1]
        gmin = (1 - w) * self.gmin[idx] + w * self.gmin[idx + 1]
        gmax = (1 - w) * self.gmax[idx] + w * self.gmax[idx + 1]

        def lerp(ga, ya, gb, yb):
            return lambda gs: (1 - w) * _interp(ga, ya, gs) + w * _interp(gb, yb, gs)

        # NB transfer-functions-2d.jl:83-84 unpacks (lower_f, upper_f, lower_t, upper_t) into variables
        # named the other way round; the (f, t) pairs stay together, so only the labels are swapped
        self._last = dict(lower_f=lerp(b1.lower_g, b1.lower_f, b2.lower_g, b2.lower_f),
                          upper_f=lerp(b1.upper_g, b1.upper_f, b2.upper_g, b2.upper_f),
                          lower_t=lerp(b1.lower_g, b1.lower_t, b2.lower_g, b2.lower_t),
                          upper_t=lerp(b1.upper_g, b1.upper_t, b2.upper_g, b2.upper_t))

        def both(gs):
            fl, fu = self._last["lower_f"](gs), self._last["upper_f"](gs)
            return np.where(np.isnan(fl), 0.0, fl) + np.where(np.isnan(fu), 0.0, fu)

        return gmin, gmax, both


def transferfunctions(m, x, d, *, minrₑ=None, maxrₑ=50.0, numrₑ=100, radii=None, h=1e-6, **kwargs):
    """transferfunctions(m, x, d; minrₑ, maxrₑ, numrₑ) (:535-556): every radius in one batch on the device."""
    from .planes import InverseGrid

    if radii is None:
        minrₑ = m.isco() + 1e-2 if minrₑ is None else minrₑ
        radii = InverseGrid()(minrₑ, maxrₑ, numrₑ)
    ctfs = cunningham_transfer_functions(m, x, d, radii, **kwargs)
    return InterpolatingTransferBranches.from_branches([interpolate_branches(c, h=h) for c in ctfs])


def _integrate_bins(S, lo, hi, gmin, gmax, h, X, W):
    """integrate_bin (integration.jl:164-203) for an array of bins [lo, hi] that overlap [gmin, gmax]:
    Gauss-Legendre inside, the closed-form edge term where 1/sqrt(g✶(1-g✶)) blows up."""
    span = gmax - gmin
    glo, ghi = np.clip(lo, gmin, gmax), np.clip(hi, gmin, gmax)
    slo, shi = (lo - gmin) / span, (hi - gmin) / span
    lum = np.zeros(lo.size)
    done = np.zeros(lo.size, dtype=bool)

    def edge(lim, lim_gs):
        gh = span * lim_gs + gmin
        return S(gh) * np.abs(np.sqrt(gh) - np.sqrt(lim)) * math.sqrt(h)

    a = slo < h
    a_in, a_all = a & (shi > h), a & ~(shi > h)
    if np.any(a_in):
        lum[a_in] += edge(glo[a_in], np.full(int(a_in.sum()), h))
        glo = np.where(a_in, span * h + gmin, glo)
    if np.any(a_all):
        lum[a_all] = edge(glo[a_all], shi[a_all])
        done |= a_all
    b = (shi > 1.0 - h) & ~done
    b_in, b_all = b & (slo < 1.0 - h), b & ~(slo < 1.0 - h)
    if np.any(b_in):
        lum[b_in] += edge(ghi[b_in], np.full(int(b_in.sum()), 1.0 - h))
        ghi = np.where(b_in, span * (1.0 - h) + gmin, ghi)
    if np.any(b_all):
        lum[b_all] = edge(ghi[b_all], slo[b_all])
        done |= b_all
    q = ~done
    if np.any(q):
        half = 0.5 * (ghi[q] - glo[q])
        nodes = (X[None, :] + 1.0) * half[:, None] + glo[q][:, None]
        vals = S(nodes.ravel()).reshape(nodes.shape)
        lum[q] += (vals @ W) * half
    return np.where(np.isfinite(lum), lum, 0.0)


def integrate_lineprofile(ε, tfs: InterpolatingTransferBranches, g_grid, *, rmin=None, rmax=None, g_scale=1.0, h=1e-8,
                          n_radii=1000, quadrature_points=7):
    """integrate_lineprofile (integration.jl:205-262,330-372): ∫∫ over rₑ (inverse grid, n_radii annuli)
    and over each bin of g (Gauss-Legendre, with the analytic treatment of the integrable
    1/sqrt(g✶(1-g✶)) edges :152-203), then `_normalize!` (utils.jl:113-125)."""
    from .planes import InverseGrid

    g_grid = np.asarray(g_grid, dtype=np.float64)
    rmin = tfs.inner_radius() if rmin is None else rmin
    rmax = tfs.outer_radius() if rmax is None else rmax
    X, W = np.polynomial.legendre.leggauss(quadrature_points)
    radii = np.asarray(InverseGrid()(rmin, rmax, n_radii))
    out = np.zeros(g_grid.size)
    lo_all, hi_all = g_grid[:-1] / g_scale, g_grid[1:] / g_scale
    r_prev = rmin - (radii[1] - rmin)
    for rₑ in radii:
        gmin, gmax, both = tfs.at(rₑ)
        span = gmax - gmin

        def S(g):
            gs = (g - gmin) / span
            with np.errstate(all="ignore"):
                return (g * g) * both(gs) * g / np.sqrt(gs * (1.0 - gs))

        θw = (rₑ - r_prev) * rₑ * ε(rₑ) * math.pi / span
        r_prev = rₑ
        glo = np.clip(lo_all, gmin, gmax)
        ghi = np.clip(hi_all, gmin, gmax)
        live = np.nonzero(glo != ghi)[0]
        if live.size == 0:
            continue
        out[live] += _integrate_bins(S, lo_all[live], hi_all[live], gmin, gmax, h, X, W) * θw
    # _normalize!
    flux = out.copy()
    flux[:-1] = flux[:-1] / (g_grid[1:] + g_grid[:-1])
    total = flux[:-1].sum()
    if total > 0:
        flux = flux / total
    return flux


def integrate_lagtransfer(prof, tfs: InterpolatingTransferBranches, g_grid, t_grid, *, rmin=None, rmax=None, g_scale=1.0,
                          h=1e-8, n_radii=1000, quadrature_points=7, t0=0.0):
    """integrate_lagtransfer (integration.jl:264-289,374-453): the (g, t) response of the disc to a flash
    of the corona.  `prof` provides emissivity_at(r) and coordtime_at(r) (source -> disc time); each
    annulus and g-bin deposits its lower- and upper-branch flux at (source -> disc) + (disc -> observer)
    - t0.  Rows are then normalised like the line profile (`_normalize!` for matrices; its final
    row-maximum rescaling is discarded by the reference's own call chain and is not applied)."""
    from .planes import GeometricGrid

    g_grid = np.asarray(g_grid, dtype=np.float64)
    t_grid = np.asarray(t_grid, dtype=np.float64)
    rmin = tfs.inner_radius() if rmin is None else rmin
    rmax = tfs.outer_radius() if rmax is None else rmax
    X, W = np.polynomial.legendre.leggauss(quadrature_points)
    radii = np.asarray(GeometricGrid()(rmin, rmax, n_radii))
    out = np.zeros((g_grid.size, t_grid.size))
    r_prev = rmin - (radii[1] - rmin)
    for rₑ in radii:
        gmin, gmax, _ = tfs.at(rₑ)
        br = tfs._last
        span = gmax - gmin

        def make_S(fb):
            def S(g):
                gs = (g - gmin) / span
                f = fb(gs)
                with np.errstate(all="ignore"):
                    return (g * g) * np.where(np.isnan(f), 0.0, f) * g / np.sqrt(gs * (1.0 - gs))
            return S

        θw = (rₑ - r_prev) * rₑ * float(prof.emissivity_at(rₑ)) * math.pi / span
        t_sd = float(prof.coordtime_at(rₑ)) - t0
        r_prev = rₑ
        glo = np.clip(g_grid[:-1] / g_scale, gmin, gmax)
        ghi = np.clip(g_grid[1:] / g_scale, gmin, gmax)
        live = np.nonzero(glo != ghi)[0]
        if live.size == 0:
            continue
        glo, ghi = glo[live], ghi[live]
        k1 = _integrate_bins(make_S(br["lower_f"]), glo, ghi, gmin, gmax, h, X, W)
        k2 = _integrate_bins(make_S(br["upper_f"]), glo, ghi, gmin, gmax, h, X, W)

        def times(gs):
            """_time_g✶ (:96-102): near an extremum the two branches' times are blended"""
            gs = np.clip(gs, 0.0, 1.0)
            tl, tu = br["lower_t"](gs), br["upper_t"](gs)
            lo_e, hi_e = gs < h, gs > 1.0 - h
            ω = np.where(lo_e, gs / h, 1.0 - (1.0 - gs) / h)
            at = np.where(lo_e, h, 1.0 - h)
            tle, tue = br["lower_t"](at), br["upper_t"](at)
            edge = lo_e | hi_e
            t1 = np.where(edge, tle * ω + (1.0 - ω) * tue, tl)
            t2 = np.where(edge, tue * ω + (1.0 - ω) * tle, tu)
            return t1, t2

        tl1, tu1 = times((glo - gmin) / span)
        tl2, tu2 = times((ghi - gmin) / span)
        i1 = np.searchsorted(t_grid, 0.5 * (tl1 + tl2) + t_sd, side="left")
        i2 = np.searchsorted(t_grid, 0.5 * (tu1 + tu2) + t_sd, side="left")
        ok1, ok2 = i1 < t_grid.size, i2 < t_grid.size
        np.add.at(out, (live[ok1], i1[ok1]), k1[ok1] * θw)
        np.add.at(out, (live[ok2], i2[ok2]), k2[ok2] * θw)
    flux = out.copy()
    flux[:-1, :] = flux[:-1, :] / (g_grid[1:] + g_grid[:-1])[:, None]
    total = flux[:-1, :].sum()
    if total > 0:
        flux = flux / total
    return flux


# ------------------------------------------------------------------------------------------
# grids and tables of transfer functions (cunningham-transfer-functions.jl:440-530, types.jl:14-130)
# ------------------------------------------------------------------------------------------
@dataclass
class CunninghamTransferGrid:
    """types.jl:14-44: the branches of every radius resampled on one g✶ grid; matrices are [g✶, r]."""

    r_grid: np.ndarray
    g_star_grid: np.ndarray
    g_min: np.ndarray
    g_max: np.ndarray
    lower_f: np.ndarray
    upper_f: np.ndarray
    lower_time: np.ndarray
    upper_time: np.ndarray

    def inner_radius(self):
        return float(self.r_grid[0])

    def outer_radius(self):
        return float(self.r_grid[-1])

    def fields(self):
        return (self.r_grid, self.g_star_grid, self.g_min, self.g_max, self.lower_f, self.upper_f, self.lower_time, self.upper_time)


def transfer_function_grid(itfs_or_metric, *args, Ng=20, h_grid=1e-3, **kwargs):
    """transfer_function_grid(itfs, Ng; h) / transfer_function_grid(m, x, d, radii; Ng, h_grid, ...) (:463-501)"""
    if isinstance(itfs_or_metric, InterpolatingTransferBranches):
        itfs = itfs_or_metric
    else:
        x, d, radii = args
        itfs = transferfunctions(itfs_or_metric, x, d, radii=radii, **kwargs)
    gs = np.linspace(0.0, 1.0, int(Ng))
    gc = np.clip(gs, h_grid, 1.0 - h_grid)
    col = lambda key_g, key_y: np.stack([_interp(getattr(b, key_g), getattr(b, key_y), gc) for b in itfs.branches], axis=1)
    return CunninghamTransferGrid(itfs.radii.copy(), gs, itfs.gmin.copy(), itfs.gmax.copy(), col("lower_g", "lower_f"),
                                  col("upper_g", "upper_f"), col("lower_g", "lower_t"), col("upper_g", "upper_t"))


@dataclass
class CunninghamTransferTable:
    """types.jl:97-130: grids on a rectangular lattice of parameters (spin, inclination); calling the table
    interpolates every field multi-linearly (the reference's MultilinearInterpolator over the same fields)."""

    params: tuple
    grids: np.ndarray          # object array, one CunninghamTransferGrid per lattice point

    def __call__(self, *point):
        axes = [np.asarray(p, dtype=np.float64) for p in self.params]
        if len(point) != len(axes):
            raise ValueError("one coordinate per table axis")
        corners = [()]
        weights = [1.0]
        for ax, p in zip(axes, point):
            if ax.size == 1:
                i, w = 0, 0.0
                lo_hi = ((i, 1.0),)
            else:
                i = int(np.clip(np.searchsorted(ax, p, side="right") - 1, 0, ax.size - 2))
                w = float((p - ax[i]) / (ax[i + 1] - ax[i]))
                lo_hi = ((i, 1.0 - w), (i + 1, w))
            corners, weights = ([c + (k,) for c in corners for k, _ in lo_hi],
                                [wt * wk for wt in weights for _, wk in lo_hi])
        acc = None
        for c, wt in zip(corners, weights):
            f = self.grids[c].fields()
            acc = [wt * a for a in f] if acc is None else [s + wt * a for s, a in zip(acc, f)]
        return CunninghamTransferGrid(*acc)


def make_transfer_function_table(metric_type, d, a_range, θ_range, *, r_max=500.0, n_radii=150, r_obs=10000.0, **kwargs):
    """make_transfer_function_table(M, d, a_range, θ_range; r_max, n_radii) (:503-530): for every (a, θ in degrees) the
    grid of `n_radii` transfer functions between isco + 1e-2 and r_max, observer at r = 10⁴.  Each lattice point is
    one batch on the device (all its radii share their launches)."""
    from .planes import InverseGrid

    a_range, θ_range = [float(a) for a in a_range], [float(t) for t in θ_range]
    grids = np.empty((len(a_range), len(θ_range)), dtype=object)
    for i, a in enumerate(a_range):
        m = metric_type(1.0, a)
        radii = InverseGrid()(m.isco() + 1e-2, r_max, n_radii)
        for j, θ in enumerate(θ_range):
            x = np.array([0.0, float(r_obs), math.radians(θ), 0.0])
            grids[i, j] = transfer_function_grid(m, x, d, radii, **kwargs)
    return CunninghamTransferTable((np.array(a_range), np.array(θ_range)), grids)
