"""Tracing front-end: configuration, charts, initial conditions and the ensemble boundary.

Mirrors src/tracing/{tracing,configuration,charts,callbacks,utility,constraints}.jl of the
reference.  Everything below `ensemble_solve_tracing_problem` runs in the HIP library; this
module only flattens the configuration into the C-ABI structs.
"""
from __future__ import annotations

import ctypes as C
import math
import dataclasses
from dataclasses import dataclass
from typing import Optional

import os

import numpy as np

from . import _lib
from .geometry import (GR_DISC_NONE, AbstractAccretionGeometry, CompositeGeometry, DatumPlane, EllipticalDisc, MeshAccretionGeometry,
                       PrecessingDisc, ShakuraSunyaev, ThickDisc, ThinDisc, WarpedThinDisc)
from .metrics import AbstractMetric
from .orthonormalization import lnrbasis

DEFAULT_TOLERANCE = 1e-9  # configuration.jl:1


# ---- charts.jl:3-6,51-58 ----
@dataclass(frozen=True)
class PolarChart:
    inner_radius: float
    outer_radius: float


def chart_for_metric(m: AbstractMetric, outer_radius: float = 12000.0, *, closest_approach: float = 1.01):
    return PolarChart(m.inner_radius() * closest_approach, outer_radius)


# ---- charts.jl:26-48,61-70 ----
@dataclass(frozen=True, eq=False)
class PoloidalShapeChart:
    """Inner boundary r_min(θ): `table[k]` at θ_k uniform on [θ_first, θ_last], interpolated linearly
    on the device (the reference wraps the same samples in a DataInterpolations.LinearInterpolation)."""

    table: np.ndarray
    θ_first: float
    θ_last: float
    outer_radius: float = 12000.0

    def shapefunc(self, θ):
        n = self.table.size
        f = (np.asarray(θ, dtype=np.float64) - self.θ_first) * ((n - 1) / (self.θ_last - self.θ_first))
        k = np.clip(np.floor(f).astype(np.int64), 0, n - 2)
        return self.table[k] + (f - k) * (self.table[k + 1] - self.table[k])


def event_horizon(m: AbstractMetric, *, select=max, resolution: int = 100, θε: float = 1e-7, rmax: float = 5.0,
                  init: float = 0.0, samples: int = 20001):
    """event_horizon(m; select = maximum, resolution, θε, rmax) special-radii.jl:105-133: for every θ
    of range(θε, 2π - θε, resolution) the roots in r of g_tϕ² - g_tt g_ϕϕ = 0 on [init, rmax], reduced
    by `select`; NaN where there is none.  Roots.find_zeros (third party) is restated as a dense
    sign-change scan followed by bisection."""
    θs = np.linspace(θε, 2.0 * math.pi - θε, resolution)
    rs = np.full(resolution, np.nan)
    rgrid = np.linspace(init, rmax, samples)[1:]           # r = 0 itself is singular
    for i, θ in enumerate(θs):
        s, c = math.sin(θ), math.cos(θ)

        def cond(r):
            g = m._components(r, s, c)
            return g[4] * g[4] - g[0] * g[3]

        with np.errstate(all="ignore"):
            f = cond(rgrid) + 0.0 * rgrid
        ok = np.isfinite(f)
        roots = [float(r) for r in rgrid[ok & (f == 0.0)]]
        idx = np.nonzero(ok[:-1] & ok[1:] & (np.signbit(f[:-1]) != np.signbit(f[1:])) & (f[:-1] != 0) & (f[1:] != 0))[0]
        for k in idx:
          with np.errstate(all="ignore"):
              lo, hi, flo = rgrid[k], rgrid[k + 1], f[k]
              for _ in range(200):
                  mid = 0.5 * (lo + hi)
                  if not (lo < mid < hi):
                      break
                  fm = cond(mid)
                  if (fm < 0) == (flo < 0):
                      lo, flo = mid, fm
                  else:
                      hi = mid
              # a sign change through a pole (1/Σ-type blow-up) is not a root
              if abs(cond(0.5 * (lo + hi))) < 1e-6 * max(1.0, np.nanmax(np.abs(f[max(k - 1, 0):k + 3]))):
                  roots.append(0.5 * (lo + hi))
        if roots:
            rs[i] = select(roots)
    return rs, θs


def is_naked_singularity(m: AbstractMetric, *, resolution: int = 100, θε: float = 1e-7, rmax: float = 5.0):
    """is_naked_singularity(m; resolution, θε, rmax) (special-radii.jl:135-146): some polar angle has no
    horizon radius in (0, rmax]."""
    rs, _ = event_horizon(m, resolution=resolution, θε=θε, rmax=rmax)
    return bool(np.any(np.isnan(rs)))


def event_horizon_chart(m: AbstractMetric, *, outer_radius: float = 12000.0, closest_approach: float = 1.01, **kwargs):
    rs, θs = event_horizon(m, **kwargs)
    return PoloidalShapeChart(np.ascontiguousarray(rs * closest_approach), float(θs[0]), float(θs[-1]), outer_radius)


# ---- tracing.jl:1-7, photon-rings.jl:17-24 ----
@dataclass(frozen=True)
class TraceGeodesic:
    """TraceGeodesic(μ = 0, q = 0)"""

    μ: float = 0.0
    q: float = 0.0


@dataclass(frozen=True)
class TraceWindings:
    """TraceWindings(μ = 0, plane_inc = π/2) -- src/tracing/photon-rings.jl: counts how often the geodesic has crossed
    the cone θ = plane_inc (checked at step ends, as the reference's DiscreteCallback does).  The count is the
    `winding` of a point (`winding_number(points)`, `ConstPointFunctions.winding()`)."""

    μ: float = 0.0
    plane_inc: float = math.pi / 2


def winding_number(points):
    """gp.aux.winding of TraceWindings end points (carried in bits 16..31 of the records' flags)."""
    return (np.asarray(points["flags"]).astype(np.uint32) >> 16).astype(np.int64)


# ---- callbacks.jl:31-40 ----
@dataclass(frozen=True)
class DomainUpperHemisphere:
    delta: float = 1e-4


def domain_upper_hemisphere(δ: float = 1e-4):
    return DomainUpperHemisphere(δ)


# ---- utility.jl:13-20 ----
def local_momentum(r_obs, α, β):
    b = β / r_obs
    a = α / r_obs
    pr = -1.0 / math.sqrt(1.0 + a * a + b * b)
    return np.array([1.0, pr, b * pr, a * pr])


# ---- utility.jl:32-40 ----
def lnr_momentum_to_global_velocity_matrix(m: AbstractMetric, x):
    """The constant 4x4 matrix of `lnr_momentum_to_global_velocity_transform`: p̄ ↦ g⁻¹ (Tx p̄)."""
    g = m.metric(x)
    Tx = np.column_stack(lnrbasis(g))
    return np.linalg.inv(g) @ Tx


def lnr_momentum_to_global_velocity_transform(m: AbstractMetric, x):
    Mx = lnr_momentum_to_global_velocity_matrix(m, x)
    return lambda pbar: Mx @ pbar


# ---- utility.jl:66-87 ----
def map_impact_parameters(m: AbstractMetric, x, α, β):
    """Unconstrained initial velocity (or an (n, 4) array of them) for impact parameters."""
    Mx = lnr_momentum_to_global_velocity_matrix(m, x)
    if np.ndim(α) == 0 and np.ndim(β) == 0:
        return Mx @ local_momentum(x[1], float(α), float(β))
    αs, βs = np.broadcast_arrays(np.asarray(α, dtype=np.float64), np.asarray(β, dtype=np.float64))
    a, b = αs.ravel() / x[1], βs.ravel() / x[1]
    pr = -1.0 / np.sqrt(1.0 + a * a + b * b)
    pbar = np.stack([np.ones_like(pr), pr, b * pr, a * pr], axis=1)          # local_momentum, row per ray
    return pbar @ Mx.T


# ---- Gradus.jl:412 / ext/GradusDiffEqGPUExt: the ensemble type selects the backend ----
class EnsembleMI355X:
    """Ensemble algorithm that runs the trace on one MI355X -- or, with `devices=[...]`, on several from this one process
    -- through libgradus_mi355x.so.

    Passing it as `ensemble=` is the drop-in point (dispatch of
    `ensemble_solve_tracing_problem`, src/tracing/tracing.jl:113-196).
    """

    def __init__(self, device: int = 0, devices=None, **knobs):
        """`devices=[0, 1, ...]`: every entry of the boundary (rendergeodesics, prerendergeodesics, tracegeodesics,
        lineprofile(BinningMethod)) spreads its rays over these GPUs through the library's *_multi entry points: the host
        enqueues every device's share, then collects them -- no exchange between devices.  An entry may repeat (two
        contexts on one device: testing)."""
        self.device = device if devices is None else list(devices)[0]
        self.devices = None if devices is None else [int(d) for d in devices]
        self.knobs = dict(knobs)
        self._ctx: Optional[_lib.Context] = None
        self._extra: list = []

    @property
    def contexts(self):
        """One context per entry of `devices` (the first is `ctx`)."""
        if self.devices is None:
            return [self.ctx]
        while len(self._extra) < len(self.devices) - 1:
            c = _lib.Context(self.devices[len(self._extra) + 1])
            for k, v in self.knobs.items():
                c.set(k, v)
            self._extra.append(c)
        return [self.ctx] + self._extra

    @property
    def ctx(self) -> _lib.Context:
        if self._ctx is None:
            self._ctx = _lib.Context(self.device)
            for k, v in self.knobs.items():
                self._ctx.set(k, v)
        return self._ctx

    @property
    def multi(self) -> bool:
        return self.devices is not None and len(self.devices) > 1

    def contexts_for_width(self, width: int):
        """The contexts an image plane of `width` columns is dealt over: all of them when their number divides the width (the
        library deals whole columns), else the largest leading subset that does -- 3 devices and a 1024-wide image render on 2
        (the Julia shim guards the same way; the library itself refuses the deal)."""
        ctxs = self.contexts
        k = len(ctxs)
        while k > 1 and width % k != 0:
            k -= 1
        return ctxs[:k]

    def set(self, key, value):
        self.knobs[key] = value
        if self._ctx is not None:
            self._ctx.set(key, value)
        for c in self._extra:
            c.set(key, value)
        return self


@dataclass(frozen=True)
class RenderVelocity:
    """`_render_velocity_function` (rendering.jl:140-163) as data: evaluated per pixel on device."""

    alpha_lims: tuple
    beta_lims: tuple
    image_width: int
    image_height: int
    offset: float = 1e-6


@dataclass(frozen=True)
class PlaneVelocity:
    """`promote_velfunc(m, x, plane::PolarPlane, _)` (image-planes/planes.jl:180-184) as data: the plane's rays are formed on
    the device from its radii and angle tables (gr_rayset.sep_*), in the order of vec(αs) -- trajectory i is
    (r index, θ index) = ((i - 1) % Nr, (i - 1) ÷ Nr)."""

    plane: object


def separable_rayset(m, position, plane, tiled: bool):
    """gr_rayset for a PolarPlane handed over as its three tables (α = r_i cos θ_j, β = r_i sin θ_j, area = r_i²,
    planes.jl:96-131).  Returns (rayset, (r, cos θ, sin θ)); the arrays must outlive the call that uses the rayset."""
    r = np.ascontiguousarray(plane.grid(plane.r_min, plane.r_max, plane.Nr), dtype=np.float64)
    dθ = (plane.θ_max - plane.θ_min) / plane.Nθ
    θs = np.linspace(plane.θ_min, plane.θ_max - dθ, plane.Nθ)
    cs, sn = np.ascontiguousarray(np.cos(θs)), np.ascontiguousarray(np.sin(θs))
    rs = _lib.gr_rayset()
    position = np.asarray(position, dtype=np.float64)
    Mx = lnr_momentum_to_global_velocity_matrix(m, position)
    for i in range(4):
        rs.x_obs[i] = float(position[i])
        for k in range(4):
            rs.Mx[4 * i + k] = float(Mx[i, k])
    rs.alpha = rs.beta = rs.area = rs.height = None
    rs.sep_r, rs.sep_cos, rs.sep_sin = r.ctypes.data, cs.ctypes.data, sn.ctypes.data
    rs.sep_nr, rs.sep_nt, rs.sep_tiled, rs.n = plane.Nr, plane.Nθ, int(bool(tiled)), plane.Nr * plane.Nθ
    return rs, (r, cs, sn)


@dataclass
class TracingConfiguration:
    """configuration.jl:3-88 flattened; `μ` from TraceGeodesic (tracing.jl:1-7), `gtol` from
    geometry/bootstrap.jl:8."""

    metric: AbstractMetric
    position: np.ndarray
    velocity: object                # (n,4) array | RenderVelocity
    geometry: Optional[AbstractAccretionGeometry]
    chart: PolarChart
    callback: object
    ensemble: EnsembleMI355X
    trajectories: Optional[int]
    λ_domain: tuple
    abstol: float = DEFAULT_TOLERANCE
    reltol: float = DEFAULT_TOLERANCE
    gtol: float = 1e-2
    μ: float = 0.0
    maxiters: int = 1_000_000
    q: float = 0.0
    winding_plane: Optional[float] = None      # TraceWindings.plane_inc (photon-rings.jl:17-24); None = plain TraceGeodesic

    def abi_config(self) -> _lib.gr_config:
        c = _lib.gr_config()
        m = self.metric
        if m.metric_id < 0:
            raise NotImplementedError(f"metric {type(m).__name__} has no device implementation")
        c.metric_id = m.metric_id
        for i, p in enumerate(m.abi_params()):
            c.params[i] = float(p)
        if hasattr(m, "table"):
            # GR_METRIC_TABULATED: the fitted table of a user-defined metric (metrics.TabulatedMetric keeps it alive).  The chart and
            # the point the rays start from must lie inside it (polynomials do not extrapolate; the library refuses): it grows first.
            ch_in = float(np.nanmin(self.chart.table)) if isinstance(self.chart, PoloidalShapeChart) else float(self.chart.inner_radius)
            r_start = float(np.asarray(self.position, dtype=np.float64).reshape(-1, 4)[:, 1].max()) if self.position is not None else ch_in
            r_start_lo = float(np.asarray(self.position, dtype=np.float64).reshape(-1, 4)[:, 1].min()) if self.position is not None else ch_in
            m.cover(min(ch_in, r_start_lo), max(float(self.chart.outer_radius), r_start))
            c.metric_table, c.metric_table_n = m.table.ctypes.data, m.table.size
        if isinstance(self.chart, PoloidalShapeChart):
            ch = self.chart
            c.r_inner, c.r_outer = float(np.nanmin(ch.table)), float(ch.outer_radius)
            c.chart_table, c.chart_table_n = ch.table.ctypes.data, ch.table.size      # the chart keeps it alive
            c.chart_theta0, c.chart_theta1 = ch.θ_first, ch.θ_last
        else:
            c.r_inner, c.r_outer = float(self.chart.inner_radius), float(self.chart.outer_radius)
        if self.geometry is None:
            c.disc_id = GR_DISC_NONE
        elif isinstance(self.geometry, CompositeGeometry):
            comps = self.geometry.geometry
            if not 2 <= len(comps) <= _lib.GR_COMP_MAX:
                raise NotImplementedError(f"a composite geometry runs on the device with 2..{_lib.GR_COMP_MAX} components")
            c.disc_id = self.geometry.disc_id
            c.comp_n = len(comps)
            for k, g in enumerate(comps):
                if not isinstance(g, (ThinDisc, ShakuraSunyaev, EllipticalDisc, DatumPlane)):
                    raise NotImplementedError(f"{type(g).__name__} cannot be a component of a composite geometry on the device")
                # every component through the single-geometry branches below, then copied into its slot
                one = dataclasses.replace(self, geometry=g).abi_config()
                c.comp[k].disc_id = one.disc_id
                c.comp[k].disc_r_in, c.comp[k].disc_r_out = one.disc_r_in, one.disc_r_out
                for q in range(4):
                    c.comp[k].disc_params[q] = one.disc_params[q]
        elif isinstance(self.geometry, ThinDisc):
            c.disc_id = self.geometry.disc_id
            c.disc_r_in, c.disc_r_out = float(self.geometry.inner_radius), float(self.geometry.outer_radius)
        elif isinstance(self.geometry, DatumPlane):
            c.disc_id = self.geometry.disc_id
            c.disc_params[0] = float(self.geometry.height)
        elif isinstance(self.geometry, EllipticalDisc):
            g = self.geometry
            c.disc_id = g.disc_id
            c.disc_r_in, c.disc_r_out = float(g.inner_radius), float("inf")
            c.disc_params[0], c.disc_params[1] = float(g.semi_major), float(g.semi_minor)
        elif isinstance(self.geometry, PrecessingDisc):
            g = self.geometry
            c.disc_id = g.disc_id
            c.disc_r_in, c.disc_r_out = float(g.disc.inner_radius), float(g.disc.outer_radius)
            c.disc_params[0], c.disc_params[1] = float(g.β), float(g.γ)
            c.disc_params[2], c.disc_params[3] = math.cos(g.β), math.sin(g.β)
        elif isinstance(self.geometry, ShakuraSunyaev):
            c.disc_id = self.geometry.disc_id
            c.disc_r_in, c.disc_r_out = float(self.geometry.inner_radius), float("inf")
            c.disc_params[0], c.disc_params[1] = float(self.geometry.Ṁ_Ṁedd), float(self.geometry.inv_η)
        elif isinstance(self.geometry, WarpedThinDisc):
            g = self.geometry
            c.disc_id = g.disc_id
            c.disc_r_in, c.disc_r_out = g.inner_radius, g.outer_radius
            c.disc_params[0], c.disc_params[1] = g.ρ_range[0], g.ρ_range[1]
            c.disc_params[2], c.disc_params[3] = float(np.abs(g.table).max()), 1.0
            c.disc_table, c.disc_table_n = g.table.ctypes.data, g.table.size     # g keeps the array alive
        elif isinstance(self.geometry, MeshAccretionGeometry):
            g = self.geometry
            c.disc_id = g.disc_id
            c.disc_table, c.disc_table_n = g.table.ctypes.data, len(g)            # g keeps the array alive
        elif isinstance(self.geometry, ThickDisc):
            g = self.geometry
            c.disc_id = g.disc_id
            c.disc_r_in, c.disc_r_out = g.inner_radius, g.outer_radius
            c.disc_params[0], c.disc_params[1], c.disc_params[2] = g.ρ_range[0], g.ρ_range[1], float(g.table.max())
            c.disc_table, c.disc_table_n = g.table.ctypes.data, g.table.size     # g keeps the array alive
        else:
            raise NotImplementedError(f"geometry {type(self.geometry).__name__} has no device implementation")
        c.gtol = float(self.gtol)
        c.lambda0, c.lambda1 = float(self.λ_domain[0]), float(self.λ_domain[1])
        c.abstol, c.reltol = float(self.abstol), float(self.reltol)
        c.mu = float(self.μ)
        c.q = float(self.q)
        if self.winding_plane is not None:
            c.count_windings, c.winding_plane = 1, float(self.winding_plane)
        c.maxiters = int(self.maxiters)
        if self.callback is None:
            c.upper_hemisphere = 0
        elif isinstance(self.callback, DomainUpperHemisphere):
            c.upper_hemisphere = 1
            c.hemi_delta = float(self.callback.delta)
        else:
            raise NotImplementedError("only `domain_upper_hemisphere()` callbacks can run on the device")
        return c

    def abi_plane(self) -> _lib.gr_plane:
        rv = self.velocity
        assert isinstance(rv, RenderVelocity)
        pl = _lib.gr_plane()
        for i in range(4):
            pl.x_obs[i] = float(self.position[i])
        Mx = lnr_momentum_to_global_velocity_matrix(self.metric, self.position)
        for i in range(4):
            for k in range(4):
                pl.Mx[4 * i + k] = float(Mx[i, k])
        pl.alpha0, pl.alpha1 = float(rv.alpha_lims[0]), float(rv.alpha_lims[1])
        pl.beta0, pl.beta1 = float(rv.beta_lims[0]), float(rv.beta_lims[1])
        pl.width, pl.height = int(rv.image_width), int(rv.image_height)
        pl.offset = float(rv.offset)
        return pl


def _as_lambda_domain(λs):
    if np.ndim(λs) == 0:
        return (0.0, float(λs))
    return (float(λs[0]), float(λs[1]))


def tracing_configuration(
    m,
    position,
    velocity,
    geometry,
    λs,
    *,
    chart=None,
    callback=None,
    ensemble=None,
    trajectories=None,
    abstol=DEFAULT_TOLERANCE,
    reltol=DEFAULT_TOLERANCE,
    gtol=1e-2,
    μ=0.0,
    maxiters=1_000_000,
    q=0.0,
    solver="Tsit5",
    save_on=False,
    trace=None,
):
    winding_plane = None
    if trace is not None:
        # TraceGeodesic(μ, q) / TraceWindings(μ, plane_inc): the `trace` keyword of tracegeodesics (tracing.jl:66-80)
        if isinstance(trace, TraceWindings):
            μ, winding_plane = trace.μ, trace.plane_inc
        elif isinstance(trace, TraceGeodesic):
            μ, q = trace.μ, trace.q
        else:
            raise NotImplementedError(f"trace {type(trace).__name__} has no device implementation")
    if solver != "Tsit5":
        raise NotImplementedError("the device integrator is Tsit5 (configuration.jl:99)")
    if q != 0.0 and getattr(m, "metric_id", None) == 11:
        # the Lorentz force on a charged test particle is a term of the Kerr-Newman kernels (kerr-newman-ad.jl:66-100): a table of
        # metric components knows nothing of the vector potential
        raise NotImplementedError("charged test particles (q != 0) are traced by KerrNewmanMetric's own kernels, not through a TabulatedMetric")
    if save_on:
        raise ValueError("Cannot use `EnsembleMI355X` with `save_on` (cf. tracing.jl:159-161)")
    if ensemble is None:
        ensemble = EnsembleMI355X()
    if not isinstance(ensemble, EnsembleMI355X):
        raise TypeError("this package only provides the `EnsembleMI355X` ensemble (no CPU fallback)")
    from .planes import AbstractImagePlane, impact_parameters

    position = np.asarray(position, dtype=np.float64)
    from .planes import PolarPlane

    if isinstance(velocity, PolarPlane) and os.environ.get("GRADUS_MI355X_SEPARABLE_RAYS", "1") != "0":
        trajectories = velocity.Nr * velocity.Nθ
        velocity = PlaneVelocity(velocity)
    elif isinstance(velocity, PlaneVelocity):
        trajectories = velocity.plane.Nr * velocity.plane.Nθ
    elif isinstance(velocity, AbstractImagePlane):
        # promote_velfunc, image-planes/planes.jl:180-184
        αs, βs = impact_parameters(velocity, position)
        velocity = map_impact_parameters(m, position, αs, βs)
        trajectories = velocity.shape[0]
    elif callable(velocity):
        # velocity function i -> SVector (1-based index, configuration.jl:47-49)
        if trajectories is None:
            raise ValueError("When velocity is a function, trajectories must be defined.")
        velocity = np.stack([np.asarray(velocity(i + 1), dtype=np.float64) for i in range(trajectories)])
    elif isinstance(velocity, RenderVelocity):
        pass
    else:
        velocity = np.ascontiguousarray(velocity, dtype=np.float64)
        if velocity.ndim == 1:
            if trajectories is not None:
                raise ValueError("Trajectories should be `nothing` when solving only a single geodesic problem.")
            velocity = velocity.reshape(1, 4)
        trajectories = velocity.shape[0]
    if chart is None:
        chart = chart_for_metric(m)
    if hasattr(geometry, "thick_disc") and not isinstance(geometry, AbstractAccretionGeometry):
        geometry = geometry.thick_disc()         # PolishDoughnut: its isobar, sampled like any ThickDisc(f)
    return TracingConfiguration(
        m, position, velocity, geometry, chart, callback, ensemble, trajectories, _as_lambda_domain(λs),
        abstol, reltol, gtol, μ, maxiters, q, winding_plane,
    )


def ensemble_solve_tracing_problem(ensemble: EnsembleMI355X, config: TracingConfiguration, *, stats: bool = False):
    """THE DROP-IN BOUNDARY (src/tracing/tracing.jl:151-196): returns the GeodesicPoint records
    (numpy structured array, 152-byte layout of src/solution-processing.jl:15-32)."""
    L = _lib.load()
    cfg = config.abi_config()
    st = _lib.gr_stats()
    if ensemble.multi:
        # several devices, one host thread: the same three input shapes through the *_multi entry points
        ctxs = ensemble.contexts
        arr, sts = _lib.ctx_array(ctxs)
        if isinstance(config.velocity, RenderVelocity):
            pl = config.abi_plane()
            ctxs = ensemble.contexts_for_width(pl.width)
            arr, sts = _lib.ctx_array(ctxs)
            n = pl.width * pl.height
            out = _lib.result_points(ensemble.ctx, n)   # pinned from 64 MiB up: every device's kernel stores its records there itself
            _lib.check(L.gr_render_endpoints_multi(arr, len(ctxs), C.byref(cfg), C.byref(pl), 0, out.ctypes.data, sts))
        elif isinstance(config.velocity, PlaneVelocity):
            rs, keep = separable_rayset(config.metric, config.position, config.velocity.plane, tiled=False)
            out = np.zeros(rs.n, dtype=_lib.POINT_DTYPE)
            _lib.check(L.gr_rayset_endpoints_multi(arr, len(ctxs), C.byref(cfg), C.byref(rs), out.ctypes.data, sts))
        else:
            v = np.ascontiguousarray(config.velocity, dtype=np.float64)
            x = np.ascontiguousarray(config.position, dtype=np.float64)
            n = v.shape[0]
            stride = 0 if x.ndim == 1 else 4
            if stride == 4 and x.shape[0] != n:
                raise ValueError("positions and velocities must have the same length")
            out = np.zeros(n, dtype=_lib.POINT_DTYPE)
            _lib.check(L.gr_trace_endpoints_multi(arr, len(ctxs), C.byref(cfg), x.ctypes.data, stride, v.ctypes.data, n,
                                                  out.ctypes.data, sts))
        st = _lib.merge_stats(sts)
        return (out, st.asdict()) if stats else out
    if isinstance(config.velocity, RenderVelocity):
        pl = config.abi_plane()
        n = pl.width * pl.height
        rg = _lib.gr_range(0, n, max(n, 1), 1)
        out = _lib.result_points(ensemble.ctx, n)       # >= 64 MiB: a block the library pinned (as the Julia shim does)
        _lib.check(L.gr_render_endpoints(ensemble.ctx.handle, C.byref(cfg), C.byref(pl), C.byref(rg),
                                         out.ctypes.data, C.byref(st)))
    elif isinstance(config.velocity, PlaneVelocity):
        # the plane's rays formed on the device, column-major like vec(αs): no (x, v) arrays to build or to send
        rs, keep = separable_rayset(config.metric, config.position, config.velocity.plane, tiled=False)
        n = rs.n
        out = np.zeros(n, dtype=_lib.POINT_DTYPE)
        _lib.check(L.gr_rayset_endpoints(ensemble.ctx.handle, C.byref(cfg), C.byref(rs), out.ctypes.data, C.byref(st)))
    else:
        v = np.ascontiguousarray(config.velocity, dtype=np.float64)
        x = np.ascontiguousarray(config.position, dtype=np.float64)
        n = v.shape[0]
        stride = 0 if x.ndim == 1 else 4
        if stride == 4 and x.shape[0] != n:
            raise ValueError("positions and velocities must have the same length")
        out = np.zeros(n, dtype=_lib.POINT_DTYPE)
        _lib.check(L.gr_trace_endpoints(ensemble.ctx.handle, C.byref(cfg), x.ctypes.data, stride, v.ctypes.data, n,
                                        out.ctypes.data, C.byref(st)))
    return (out, st.asdict()) if stats else out


def tracegeodesics(m, x, v, *args, stats=False, **kwargs):
    """tracegeodesics(m, x, v, [disc], λ_domain; kwargs...) -- src/tracing/tracing.jl:66-80.

    `v` may be an (n, 4) array of unconstrained velocities (with `x` one position or an (n, 4)
    array), an `AbstractImagePlane`, or a function `i -> velocity` with `trajectories=`.
    Returns an array of GeodesicPoint records (the reference's `EnsembleEndpointThreads` result).
    """
    if len(args) == 2:
        geometry, λs = args
    elif len(args) == 1:
        geometry, λs = None, args[0]
    else:
        raise TypeError("tracegeodesics(m, x, v, [disc], λ_domain; ...)")
    config = tracing_configuration(m, x, v, geometry, λs, **kwargs)
    return ensemble_solve_tracing_problem(config.ensemble, config, stats=stats)


@dataclass
class GeodesicPath:
    """Every accepted step of one geodesic (the reference's single-problem ODESolution)."""

    λ: np.ndarray        # (n,)
    x: np.ndarray        # (n, 4)
    v: np.ndarray        # (n, 4)
    point: np.ndarray    # GeodesicPoint record of the end point


def tracegeodesic_path(m, x, v, *args, cap=200_000, **kwargs):
    """tracegeodesics(m, x::SVector, v::SVector, [disc], λ_domain; μ, chart, ...) for ONE geodesic,
    returning the saved path (src/tracing/tracing.jl:88-110).  Runs on the device (gr_trace_path)."""
    if len(args) == 2:
        geometry, λs = args
    elif len(args) == 1:
        geometry, λs = None, args[0]
    else:
        raise TypeError("tracegeodesic_path(m, x, v, [disc], λ_domain; ...)")
    v = np.ascontiguousarray(v, dtype=np.float64).reshape(4)
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(4)
    config = tracing_configuration(m, x, v, geometry, λs, **kwargs)
    cfg = config.abi_config()
    L = _lib.load()
    path = np.zeros((cap, 9))
    n = C.c_int64(0)
    pt = np.zeros(1, dtype=_lib.POINT_DTYPE)
    _lib.check(L.gr_trace_path(config.ensemble.ctx.handle, C.byref(cfg), x.ctypes.data, v.ctypes.data, cap,
                               path.ctypes.data, C.byref(n), pt.ctypes.data))
    rows = min(n.value, cap)
    if n.value > cap:
        raise RuntimeError(f"path needs {n.value} rows; raise cap")
    path = path[:rows]
    return GeodesicPath(path[:, 0].copy(), path[:, 1:5].copy(), path[:, 5:9].copy(), pt[0])


def tracegeodesic_paths(m, x, v, *args, cap=2048, **kwargs):
    """tracegeodesics(m, xs, vs, [disc], λ_domain; ...) with save_on = true for MANY geodesics: every
    accepted step of every ray (src/tracing/tracing.jl:113-149), one ray per lane on the device
    (gr_trace_paths).  `x` is one position or an (n, 4) array; returns a list of GeodesicPath."""
    if len(args) == 2:
        geometry, λs = args
    elif len(args) == 1:
        geometry, λs = None, args[0]
    else:
        raise TypeError("tracegeodesic_paths(m, x, v, [disc], λ_domain; ...)")
    v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 4)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = v.shape[0]
    stride = 0 if x.ndim == 1 else 4
    if stride == 4 and x.shape[0] != n:
        raise ValueError("positions and velocities must have the same length")
    config = tracing_configuration(m, x if stride == 0 else x[0], v, geometry, λs, **kwargs)
    cfg = config.abi_config()
    L = _lib.load()
    while True:
        path = np.zeros((n, cap, 9))
        rows = np.zeros(n, dtype=np.int64)
        pts = np.zeros(n, dtype=_lib.POINT_DTYPE)
        _lib.check(L.gr_trace_paths(config.ensemble.ctx.handle, C.byref(cfg), x.ctypes.data, stride, v.ctypes.data, n, cap,
                                    path.ctypes.data, rows.ctypes.data, pts.ctypes.data))
        if n == 0 or rows.max() <= cap:
            break
        cap = int(rows.max())                      # a ray needed more rows than provided: once more, large enough
    return [GeodesicPath(path[j, :rows[j], 0].copy(), path[j, :rows[j], 1:5].copy(), path[j, :rows[j], 5:9].copy(), pts[j])
            for j in range(n)]
