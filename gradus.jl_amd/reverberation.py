"""Two-dimensional (time, energy) transfer functions by binning: `lagtransfer`, `binflux`
(src/transfer-functions/transfer-functions-2d.jl:86-243, src/corona/analytic.jl).

Both ray sets -- corona -> disc and observer -> disc -- are traced on the device through
`gr_trace_endpoints`; the reduction below is vectorised numpy on the end points."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import corona as K
from .planes import GeometricGrid, PolarPlane, unnormalized_areas
from .status import StatusCodes
from .tracing import chart_for_metric, domain_upper_hemisphere, tracegeodesics


@dataclass
class LagTransferFunction:
    """types.jl:156-162"""

    max_t: float
    x: np.ndarray
    image_plane_areas: np.ndarray
    coronal_geodesics: K.CoronaGeodesics
    observer_to_disc: np.ndarray


class AnalyticRadialDiscProfile:
    """AnalyticRadialDiscProfile(emissivity, cg) (analytic.jl:1-37): an analytic ε(r) with the
    source-to-disc coordinate time interpolated over the corona's disc hits."""

    def __init__(self, emissivity, cg: K.CoronaGeodesics):
        ρ = K._equatorial_project(cg.geodesic_points["x"])
        J = np.argsort(ρ, kind="stable")
        self.ε = emissivity
        self.radii = ρ[J]
        self.times = cg.geodesic_points["x"][J, 0]

    def emissivity_at(self, r):
        return self.ε(np.asarray(r, dtype=np.float64))

    def coordtime_at(self, r):
        r = np.clip(np.asarray(r, dtype=np.float64), self.radii[0], self.radii[-1])
        return K._nan_linear_interp(self.radii, self.times, r)


def observer_to_disc(m, u, plane, d, max_t, **solver_opts):
    """transfer-functions-2d.jl:137-155"""
    return tracegeodesics(m, u, plane, d, (0.0, max_t), **solver_opts)


def lagtransfer(m, u, d, model, *, plane=None, max_t=None, n_samples=10_000, sampler=None, ensemble=None, **solver_opts):
    """lagtransfer(m, u, d, model; plane, max_t, n_samples, sampler) (:157-209)"""
    u = np.asarray(u, dtype=np.float64)
    plane = PolarPlane(GeometricGrid(), Nr=800, Nθ=800, r_max=50.0) if plane is None else plane
    max_t = 2.0 * u[1] if max_t is None else max_t
    sampler = K.EvenSampler(K.BothHemispheres(), K.RandomGenerator()) if sampler is None else sampler
    solver_opts.pop("callback", None)          # both traces use domain_upper_hemisphere() (:176,:196)
    ce = K.tracecorona(m, d, model, λmax=max_t, n_samples=n_samples, sampler=sampler, ensemble=ensemble, **solver_opts)
    o_to_d = observer_to_disc(m, u, plane, d, max_t, chart=chart_for_metric(m, 1.1 * u[1]),
                              callback=domain_upper_hemisphere(), ensemble=ensemble, **solver_opts)
    return assemble_lagtransfer(max_t, u, plane, ce, o_to_d)


def assemble_lagtransfer(max_t, u, plane, ce, o_to_d):
    I = o_to_d["status"] == StatusCodes.IntersectedWithGeometry
    areas = unnormalized_areas(plane).ravel(order="F")[I]
    return LagTransferFunction(float(max_t), np.asarray(u, dtype=np.float64), areas, ce, o_to_d[I])


def _bucket_index(values, bins):
    return np.clip(np.searchsorted(bins, values, side="right") - 1, 0, bins.size - 1)


def bin_transfer_function(time_delays, energy, flux, *, N_E=300, N_t=300, energy_lims=None, time_lims=None):
    """bin_transfer_function (:98-121): Σ flux per (energy, time) cell / (ΔE Δt); empty cells NaN.
    Returns (time_bins, energy_bins, matrix[N_E, N_t])."""
    energy_lims = (float(np.min(energy)), float(np.max(energy))) if energy_lims is None else energy_lims
    time_lims = (float(np.min(time_delays)), float(np.max(time_delays))) if time_lims is None else time_lims
    eb = np.linspace(energy_lims[0], energy_lims[1], N_E)
    tb = np.linspace(time_lims[0], time_lims[1], N_t)
    de, dt = eb[1] - eb[0], tb[1] - tb[0]
    ie, it = _bucket_index(energy, eb), _bucket_index(time_delays, tb)
    tf = np.zeros((N_E, N_t))
    np.add.at(tf, (ie, it), flux)
    tf = tf / (de * dt)
    tf[tf == 0.0] = np.nan
    return tb, eb, tf


def binflux(tf: LagTransferFunction, profile=None, *, redshift=None, g=None, E0=6.4, t0=None, ensemble=None, **kwargs):
    """binflux(tf, [profile]; redshift, E₀, t0, N_t, N_E) (:211-241).  `g` may be given directly (the CPU
    tests evaluate the redshift with the oracle); by default it is evaluated on the device."""
    profile = AnalyticRadialDiscProfile(lambda r: r ** -3.0, tf.coronal_geodesics) if profile is None else profile
    t0 = tf.x[1] if t0 is None else t0
    pts = tf.observer_to_disc
    ρ = K._equatorial_project(pts["x"])
    t = profile.coordtime_at(ρ) + pts["x"][:, 0]
    ε = profile.emissivity_at(ρ)
    if g is None:
        from .pointfunctions import ConstPointFunctions
        from .rendering import apply_pointfunction
        from .tracing import tracing_configuration

        m = tf.coronal_geodesics.metric
        redshift = ConstPointFunctions.redshift(m, tf.x, **({"ensemble": ensemble} if m.metric_id != 0 else {})) \
            if redshift is None else redshift
        config = tracing_configuration(m, tf.x, np.zeros((1, 4)), tf.coronal_geodesics.geometry, tf.max_t, ensemble=ensemble)
        g = apply_pointfunction(config.ensemble, config, redshift, pts, tf.max_t)
    f = g ** 3 * ε * tf.image_plane_areas
    F = f / f.sum()
    tb, eb, td = bin_transfer_function(t, g * E0, F, **kwargs)
    return tb - t0, eb, td


# ------------------------------------------------------------------------------------------
# lag-frequency spectra (src/reverberation.jl:1-45)
# ------------------------------------------------------------------------------------------
def extend_domain_with_zeros(x, y, x_max):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    dx = x[1] - x[0]
    n = int(math.floor((x_max - x.min()) / dx + 1e-9)) + 1          # range(minimum(x), x_max, step = Δx)
    xb = x.min() + dx * np.arange(n)
    yb = np.zeros(n)
    yb[:y.size] = y
    return xb, yb


def sum_impulse_response(f):
    return np.nansum(np.asarray(f, dtype=np.float64), axis=0)


def lag_frequency(t, ψ, *, R=1.0, flo=5e-5):
    """lag_frequency(t, ψ) / lag_frequency(t, f::Matrix; flo) (reverberation.jl:29-45): time lag
    -atan(Im F / (1 + Re F)) / (2π ν) of the impulse response's Fourier transform at ν > 0.
    A matrix (g, t) is first summed over g and zero-padded to 1 / flo."""
    ψ = np.asarray(ψ, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    if ψ.ndim == 2:
        t, ψ = extend_domain_with_zeros(t, sum_impulse_response(ψ), 1.0 / flo)
    n = t.size
    freq = np.fft.fftfreq(n, d=(t[1] - t[0]))
    F = R * np.fft.fft(ψ)
    I = slice(0, n // 2)
    with np.errstate(all="ignore"):
        φ = np.arctan(F.imag[I] / (1.0 + F.real[I]))
        τ = φ / (2.0 * math.pi * freq[I])
    return freq[I], -τ


# ------------------------------------------------------------------------------------------
# continuum (source -> observer) light-travel time, src/reverberation.jl:81-93
# ------------------------------------------------------------------------------------------
def continuum_time(m, x, model, *, ensemble=None, tracer=None, tol=1e-7, max_iter=40, **solver_opts):
    """Coordinate time of the geodesic joining the observer `x` and the (point) source of `model`.

    The reference minimises, with Nelder-Mead over (α, β), the closest approach of the observer's ray
    to the source (a custom ContinuousCallback records the distance and stops the ray within 1e-2 of
    it) and reads the time where that ray stopped.  Here the same ray is found as a root instead: the
    observer's rays are traced against the horizontal plane through the source (`DatumPlane(z_src)`,
    end points only, so the whole search runs through `gr_trace_endpoints`), and a Newton iteration
    on (α, β) -- three rays per launch, Jacobian by differences -- moves the crossing point onto the
    source.  The time of that crossing is the answer; it differs from the reference's by at most its
    1e-2 stopping radius."""
    from .geometry import DatumPlane
    from .tracing import map_impact_parameters

    x = np.asarray(x, dtype=np.float64)
    pos, _ = model.sample_position_velocity(m)
    rs, θs, ϕs = pos[1], pos[2], pos[3]
    target = np.array([rs * math.sin(θs) * math.cos(ϕs), rs * math.sin(θs) * math.sin(ϕs)])
    plane = DatumPlane(rs * math.cos(θs))
    max_t = 2.0 * x[1]
    if tracer is None:
        chart = chart_for_metric(m, 2.0 * x[1])

        def tracer(α, β):
            v = map_impact_parameters(m, x, np.asarray(α, dtype=np.float64), np.asarray(β, dtype=np.float64))
            return tracegeodesics(m, x, v, plane, max_t, chart=chart, ensemble=ensemble, **solver_opts)

    def hit_xy(pts):
        r, θ, ϕ = pts["x"][:, 1], pts["x"][:, 2], pts["x"][:, 3]
        return np.stack([r * np.sin(θ) * np.cos(ϕ), r * np.sin(θ) * np.sin(ϕ)], axis=1)

    p = np.array([0.0, 0.0])
    δ = 1e-4 * max(1.0, rs)
    for _ in range(max_iter):
        pts = tracer(np.array([p[0], p[0] + δ, p[0]]), np.array([p[1], p[1], p[1] + δ]))
        if not np.all(pts["status"] == StatusCodes.IntersectedWithGeometry):
            raise RuntimeError("continuum_time: the observer's rays do not reach the source plane")
        xy = hit_xy(pts)
        F = xy[0] - target
        if math.hypot(*F) <= tol * max(1.0, rs):
            return float(pts["x"][0, 0])
        J = np.column_stack([(xy[1] - xy[0]) / δ, (xy[2] - xy[0]) / δ])
        step = np.linalg.solve(J, F)
        # damp steps that would leave the neighbourhood in which the plane is still hit
        lim = 0.5 * max(5.0, float(np.hypot(*p)) + rs)
        n = float(np.hypot(*step))
        if n > lim:
            step *= lim / n
        p = p - step
    raise RuntimeError("continuum_time did not converge")
