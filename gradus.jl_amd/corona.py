"""Corona -> disc tracing and emissivity profiles (first half).

Host-side mirror of src/corona/{corona-models,samplers,emissivity,flux-calculations,radial,spectra}.jl
and src/corona/models/lamp-post.jl.  Every geodesic is traced by the device integrator through
`tracegeodesics(m, xs, vs, d, λmax; callback = domain_upper_hemisphere())` -- the same
`gr_trace_endpoints` entry point as any other array input; what is here is the source-frame set-up
(sky angles -> tetrad -> global velocity) and the reduction of the end points to a radial
emissivity profile.  No tracing happens on the host.

`bucket!(IndexBucket, Simple(), radii, bins)` is Buckets.jl (third party): restated, as in
lineprofiles.py, as "last bin edge <= value, clamped to the first / last bin".  The reference's
golden emissivity vector (test/unit/emissivity.jl:27-48, N = 10 bins, asserted at rtol 1e-2) pins
that choice: it is reproduced to 1e-10 with this rule.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .orthonormalization import tetradframe, tetradframe_batch
from .planes import GeometricGrid
from .status import StatusCodes
from .tracing import domain_upper_hemisphere, tracegeodesics as _tracegeodesics

TWO_PI = 2.0 * math.pi


# ------------------------------------------------------------------------------------------
# samplers (src/corona/samplers.jl)
# ------------------------------------------------------------------------------------------
class LowerHemisphere:
    pass


class BothHemispheres:
    pass


class RandomGenerator:
    """geti = rand()·N (samplers.jl:32-33); `seed` makes a run repeatable (the reference draws from
    Julia's global RNG)."""

    def __init__(self, seed=None):
        self.rng = np.random.default_rng(seed)


class GoldenSpiralGenerator:
    pass


class EvenGenerator:
    pass


class EvenSampler:
    def __init__(self, domain=None, generator=None):
        self.domain = LowerHemisphere() if domain is None else domain
        self.generator = GoldenSpiralGenerator() if generator is None else generator


class WeierstrassSampler:
    def __init__(self, res=100.0, domain=None, generator=None):
        self.resolution = float(res)
        self.domain = LowerHemisphere() if domain is None else domain
        self.generator = GoldenSpiralGenerator() if generator is None else generator


def geti(sampler, index, N):
    """samplers.jl:30-33; `index` is the 1-based sample number (array or scalar)."""
    index = np.asarray(index, dtype=np.float64)
    g = sampler.generator
    if isinstance(g, EvenGenerator):
        return index / N
    if isinstance(g, GoldenSpiralGenerator):
        return index
    if isinstance(g, RandomGenerator):
        return g.rng.random(index.shape) * N
    raise NotImplementedError(type(g).__name__)


def sample_radial(sampler, i):
    if isinstance(sampler.generator, GoldenSpiralGenerator):
        return math.pi * (1.0 + math.sqrt(5.0)) * i
    return TWO_PI * i


def sample_elevation(sampler, i):
    both = isinstance(sampler.domain, BothHemispheres)
    if isinstance(sampler, EvenSampler):
        return np.arccos(1.0 - 2.0 * i) if both else np.arccos(1.0 - i)
    if isinstance(sampler, WeierstrassSampler):
        ph = 2.0 * np.arctan(np.sqrt(sampler.resolution / i))
        if not both:
            return ph
        even = (np.floor(i) == i) & (np.mod(i, 2.0) == 0.0)      # iseven on a Float64
        return np.where(even, ph, math.pi - ph)
    raise NotImplementedError(f"Not implemented for {type(sampler).__name__}.")


def sample_angles(sampler, i, N):
    """(θ, ϕ) on the emitter's sky (samplers.jl:41-44)."""
    i = np.asarray(i, dtype=np.float64)
    el = sample_elevation(sampler, i) if isinstance(sampler, WeierstrassSampler) else sample_elevation(sampler, i / N)
    return el, np.mod(sample_radial(sampler, i), TWO_PI)


def _cart_to_spher_jacobian(θ, ϕ):
    s, c, sp, cp = float(np.sin(θ)), float(np.cos(θ)), float(np.sin(ϕ)), float(np.cos(ϕ))      # (numpy's: as _sky_rows_batch)
    return np.array([[s * cp, s * sp, c], [c * cp, c * sp, -s], [-sp, cp, 0.0]])


def _cart_local_direction(θ, ϕ):
    θ, ϕ = np.asarray(θ, dtype=np.float64), np.asarray(ϕ, dtype=np.float64)
    return np.stack([np.sin(θ) * np.cos(ϕ), np.sin(θ) * np.sin(ϕ), np.cos(θ) + 0.0 * ϕ], axis=-1)


def _components_at(m, x):
    """the five metric components at x with numpy's sin / cos (what the array routes below use: the same bits either way)"""
    return tuple(float(q) for q in m._components(x[1], float(np.sin(x[2])), float(np.cos(x[2]))))


def _metric_matrix(g):
    G4 = np.zeros((4, 4))
    G4[0, 0], G4[1, 1], G4[2, 2], G4[3, 3] = g[0], g[1], g[2], g[3]
    G4[0, 3] = G4[3, 0] = g[4]
    return G4


def tetradframe_matrix(m, x, v):
    return np.column_stack(tetradframe(_metric_matrix(_components_at(m, x)), v))


def sky_angles_to_velocity(m, x, v_source, θ, ϕ, E0=1.0):
    """samplers.jl:81-99.  θ, ϕ may be arrays: returns (n, 4) unconstrained velocities."""
    hat = -_cart_local_direction(θ, ϕ)
    k = hat @ _cart_to_spher_jacobian(x[2], x[3]).T
    p = np.concatenate([np.full(k.shape[:-1] + (1,), E0), E0 * k], axis=-1)
    return p @ tetradframe_matrix(m, x, v_source).T


# ------------------------------------------------------------------------------------------
# corona models (src/corona/models/lamp-post.jl)
# ------------------------------------------------------------------------------------------
class AbstractCoronaModel:
    point_source = False      # on-axis point source: the angular emissivity method applies
    fixed_position = False    # sample_position_velocity is deterministic: one tetrad serves every sample

    def sample_position_velocity(self, m):
        raise NotImplementedError(
            f"This functions needs to be implemented for {type(self).__name__}. See the documentation for this "
            "function for instructions.")


@dataclass(frozen=True)
class LampPostModel(AbstractCoronaModel):
    h: float = 5.0
    θ: float = 0.01
    ϕ: float = 0.0
    point_source = True

    def sample_position_velocity(self, m):
        x = np.array([0.0, self.h, self.θ, self.ϕ])
        g = m.metric_components(x[1], x[2])
        return x, np.array([1.0 / math.sqrt(-g[0]), 0.0, 0.0, 0.0])


def constrain_time(g, v, μ=0.0):
    """constrain_time (auto-diff.jl:161-179) on the host, for source velocities."""
    disc = -g[0] * g[1] * v[1] ** 2 - g[0] * g[2] * v[2] ** 2 - g[0] * μ * μ - (g[0] * g[3] - g[4] ** 2) * v[3] ** 2
    return -(g[4] * v[3] + math.sqrt(disc)) / g[0]


def _dot(g, a, b):
    """g_{μν} a^μ b^ν from the five components; works on (..., 4) arrays."""
    return (g[0] * a[..., 0] * b[..., 0] + g[1] * a[..., 1] * b[..., 1] + g[2] * a[..., 2] * b[..., 2]
            + g[3] * a[..., 3] * b[..., 3] + g[4] * (a[..., 0] * b[..., 3] + a[..., 3] * b[..., 0]))


def constrain_normalize(m, x, v, μ=0.0):
    """constraints.jl:27-30"""
    g = m.metric_components(x[1], x[2])
    v = np.asarray(v, dtype=np.float64)
    vn = v / math.sqrt(abs(_dot(g, v, v)))
    return np.array([constrain_time(g, vn, μ), vn[1], vn[2], vn[3]])


@dataclass(frozen=True)
class BeamedPointSource(AbstractCoronaModel):
    """Point source on the axis moving radially with speed β (lamp-post.jl:26-48)."""

    r: float
    β: float
    point_source = True

    def sample_position_velocity(self, m):
        x = np.array([0.0, self.r, 1e-4, 0.0])
        g = m.metric_components(x[1], x[2])
        vbar = np.array([1.0, self.β * math.sqrt(-g[0] / g[1]), 0.0, 0.0])
        return x, constrain_normalize(m, x, vbar, μ=1.0)


class SourceVelocities:
    """src/corona/models/extended.jl:1-46"""

    @staticmethod
    def co_rotating(m, x):
        """the source co-rotates with the (Keplerian) disc below it"""
        s = float(np.sin(x[2]))
        v = circular_fourvelocity(m, np.array([max(m.isco(), x[1] * s)]))[0] * s
        g = _components_at(m, x)
        v = v / math.sqrt(abs(_dot(g, v, v)))
        return np.array([constrain_time(g, v, 1.0), v[1], v[2], v[3]])

    @staticmethod
    def stationary(m, x):
        g = _components_at(m, x)
        return np.array([1.0 / math.sqrt(-g[0]), 0.0, 0.0, 0.0])

    @staticmethod
    def batch(vf, m, x):
        """vf(m, x[k]) for every row of x (n, 4), or None for a velocity function this module does not know (a user's callable: the
        caller then asks sample by sample)."""
        x = np.asarray(x, dtype=np.float64)
        s, c = np.sin(x[:, 2]), np.cos(x[:, 2])
        g = [np.asarray(q, dtype=np.float64) + 0.0 * x[:, 1] for q in m._components(x[:, 1], s, c)]
        out = np.zeros_like(x)
        if vf is SourceVelocities.stationary:
            out[:, 0] = 1.0 / np.sqrt(-g[0])
            return out
        if vf is SourceVelocities.co_rotating:
            v = circular_fourvelocity(m, np.maximum(m.isco(), x[:, 1] * s)) * s[:, None]
            v = v / np.sqrt(np.abs(_dot(g, v, v)))[:, None]
            disc = -g[0] * g[1] * v[:, 1] ** 2 - g[0] * g[2] * v[:, 2] ** 2 - g[0] - (g[0] * g[3] - g[4] ** 2) * v[:, 3] ** 2      # constrain_time, μ = 1
            out[:, 0] = -(g[4] * v[:, 3] + np.sqrt(disc)) / g[0]
            out[:, 1:] = v[:, 1:]
            return out
        return None


class RingCorona(AbstractCoronaModel):
    """RingCorona(vf, r, h): an infinitely thin ring of radius r at height h (extended.jl:55-82).  A
    representative point of the ring is returned; axis symmetry does the rest.  Its emissivity goes
    through the generic Monte-Carlo route (`emissivity_profile(..., sampler=...)`); the reference's
    dedicated arm-by-arm integrator (ring.jl) is not restated."""

    fixed_position = True

    def __init__(self, *args, r=5.0, h=5.0, vf=SourceVelocities.co_rotating):
        # RingCorona(vf, r, h) | RingCorona(r, h) | RingCorona(; r, h, vf)   (extended.jl:67-71)
        if len(args) == 3:
            vf, r, h = args
        elif len(args) == 2:
            r, h = args
        elif args:
            raise TypeError("RingCorona(vf, r, h) | RingCorona(r, h) | RingCorona(r=, h=, vf=)")
        self.vf, self.r, self.h = vf, float(r), float(h)

    def sample_position_velocity(self, m):
        x = np.array([0.0, math.hypot(self.r, self.h), math.atan2(self.r, self.h), 0.0])
        return x, np.asarray(self.vf(m, x), dtype=np.float64)


class DiscCorona(AbstractCoronaModel):
    """DiscCorona(vf, r, h): a disc of radius r at height h above the accretion disc (extended.jl:165-181); every sample leaves
    from a point of its own, x = rand() r along the disc (sample_position_velocity, extended.jl:176-183).  Its emissivity goes
    through the Monte-Carlo route (`emissivity_profile(..., sampler=...)`: tracecorona + RadialDiscProfile); the reference's
    concentric-ring method (extended.jl:185-200, time-dependent ring profiles) is not restated.  `seed`: the generator behind rand()."""

    def __init__(self, *args, r=5.0, h=5.0, vf=SourceVelocities.co_rotating, seed=None):
        if len(args) == 3:
            vf, r, h = args
        elif len(args) == 2:
            r, h = args
        elif args:
            raise TypeError("DiscCorona(vf, r, h) | DiscCorona(r, h) | DiscCorona(r=, h=, vf=)")
        self.vf, self.r, self.h = vf, float(r), float(h)
        self.rng = np.random.default_rng(seed)

    def sample_position_velocity(self, m):
        ρ = self.rng.random() * self.r
        # (numpy's hypot / arctan2, as sample_positions: the two routes give the same bits)
        x = np.array([0.0, float(np.hypot(ρ, self.h)), float(np.arctan2(ρ, self.h)), 0.0])      # (x, y) flipped: off the z axis
        return x, np.asarray(self.vf(m, x), dtype=np.float64)

    def sample_positions(self, m, n, r_reject=0.0):
        """The positions of the next n calls of sample_position_velocity -- the same draws in the same order, a draw inside `r_reject`
        redrawn as corona-models.jl:1-33 does -- as an (n, 4) array."""
        xs = np.zeros((0, 4))
        if math.hypot(self.r, self.h) < r_reject:
            raise ValueError("every position of the source lies inside 1.9 inner radii")
        while xs.shape[0] < n:
            ρ = self.rng.random(n - xs.shape[0]) * self.r
            x = np.stack([np.zeros_like(ρ), np.hypot(ρ, self.h), np.arctan2(ρ, self.h), np.zeros_like(ρ)], axis=1)
            xs = np.concatenate([xs, x[x[:, 1] >= r_reject]])
        return xs


def oblate_spheroid_to_spherical(x, h, a):
    """utils.jl:186-200: (x along the equatorial axis, height h) -> Boyer-Lindquist (r, θ)"""
    if abs(a) < 1e-8:
        return math.hypot(x, h), math.atan2(x, h)
    cosθ = math.sqrt((math.sqrt(4 * a * a * h * h + (h * h + x * x - a * a) ** 2) + a * a - h * h - x * x) / (2 * a * a))
    return h / cosθ, math.acos(cosθ)


def sample_position_direction_velocity(m, model, sampler, N):
    """corona-models.jl:1-33 -> (xs, vs, vs_source), each (N, 4)."""
    idx = np.arange(1, N + 1)
    i = geti(sampler, idx, N)
    θ, ϕ = sample_angles(sampler, i, N)
    rmin = m.inner_radius() * 1.9
    if model.point_source or model.fixed_position:
        x, v = model.sample_position_velocity(m)
        if x[1] < rmin:
            raise ValueError("source position lies inside 1.9 inner radii")
        x = x.copy()
        x[2] = min(max(x[2], 1e-3), math.pi - 1e-3)             # avoid coordinate singularities :18-24
        vs = sky_angles_to_velocity(m, x, v, θ, ϕ)
        return np.tile(x, (N, 1)), vs, np.tile(v, (N, 1))
    batch = _sky_rows_batch(m, model, int(N), rmin) if hasattr(model, "sample_positions") else None
    if batch is not None:
        # every sample's position and matrix at once (the rows the device route hands over): v = M (1, -k̂)
        rows, vsrc = batch
        hat = -_cart_local_direction(θ, ϕ)
        pb = np.concatenate([np.ones((int(N), 1)), hat], axis=1)
        return rows[:, 0:4].copy(), np.einsum("nij,nj->ni", rows[:, 4:20].reshape(int(N), 4, 4), pb), vsrc
    xs, vs, vsrc = np.zeros((N, 4)), np.zeros((N, 4)), np.zeros((N, 4))
    for k in range(N):
        x, v = model.sample_position_velocity(m)
        redraws = 0
        while x[1] < rmin:
            x, v = model.sample_position_velocity(m)
            redraws += 1
            if redraws > 100_000:
                raise ValueError("source positions lie inside 1.9 inner radii")
        x = np.array(x, dtype=np.float64)
        x[2] = min(max(x[2], 1e-3), math.pi - 1e-3)
        xs[k], vsrc[k] = x, v
        vs[k] = sky_angles_to_velocity(m, x, v, θ[k], ϕ[k])
    return xs, vs, vsrc


def _device_corona_enabled(m, sampler):
    """GRADUS_MI355X_DEVICE_CORONA=0 keeps the record route (tracecorona + build_radial_profile on the host's numpy)."""
    import os

    return (os.environ.get("GRADUS_MI355X_DEVICE_CORONA", "1") != "0" and isinstance(sampler, (EvenSampler, WeierstrassSampler))
            and getattr(m, "metric_id", None) is not None)


def _sky_route(m, model, sampler):
    """Sky rays formed on the device: a source at one position crosses the boundary as one matrix, any other as 28 doubles per
    sample (gr_rayset.sky_rows, ABI 8)."""
    return _device_corona_enabled(m, sampler)


def _sky_endpoints(m, model, sampler, n_samples, geometry, λs, stats=False, **kwargs):
    """End-point records of a sky ray set (gr_rayset_endpoints): the directions are formed per lane on the device, nothing
    per ray crosses PCIe on the way in (a caller's generator: its 8 bytes per ray)."""
    import ctypes as C

    from . import _lib
    from .tracing import tracing_configuration

    rs, keep, x, v_src = sky_rayset(m, model, sampler, n_samples)
    config = tracing_configuration(m, x, np.zeros((1, 4)), geometry, λs, **kwargs)
    cfg = config.abi_config()
    out = np.zeros(rs.n, dtype=_lib.POINT_DTYPE)
    st = _lib.gr_stats() if stats else None
    _lib.check(_lib.load().gr_rayset_endpoints(config.ensemble.ctx.handle, C.byref(cfg), C.byref(rs), out.ctypes.data,
                                               C.byref(st) if stats else None))
    return (out, st, v_src) if stats else (out, v_src)


def tracegeodesics(m, model, *args, n_samples=1024, sampler=None, **kwargs):
    """tracegeodesics(m, model::AbstractCoronaModel, [d], λ; n_samples, sampler) corona-models.jl:143-153"""
    sampler = EvenSampler(BothHemispheres(), GoldenSpiralGenerator()) if sampler is None else sampler
    if _sky_route(m, model, sampler) and len(args) in (1, 2) and not kwargs.get("stats"):
        geometry, λs = args if len(args) == 2 else (None, args[0])
        return _sky_endpoints(m, model, sampler, n_samples, geometry, λs, **kwargs)[0]
    xs, vs, _ = sample_position_direction_velocity(m, model, sampler, n_samples)
    return _tracegeodesics(m, xs, vs, *args, **kwargs)


@dataclass
class CoronaGeodesics:
    metric: object
    geometry: object
    model: object
    geodesic_points: np.ndarray
    source_velocity: np.ndarray


def tracecorona(m, g, model, *, λmax=10_000.0, n_samples=1024, sampler=None, callback="default", **kwargs):
    """corona-models.jl:164-190"""
    sampler = EvenSampler(BothHemispheres(), RandomGenerator()) if sampler is None else sampler
    if callback == "default":
        callback = domain_upper_hemisphere()
    if _sky_route(m, model, sampler):
        gps, v_src = _sky_endpoints(m, model, sampler, n_samples, g, λmax, callback=callback, **kwargs)
        mask = gps["status"] == StatusCodes.IntersectedWithGeometry
        # (a source of many positions: v_src holds one source velocity per sample)
        return CoronaGeodesics(m, g, model, gps[mask], v_src[mask] if np.ndim(v_src) == 2 else np.tile(v_src, (int(mask.sum()), 1)))
    xs, vs, vsrc = sample_position_direction_velocity(m, model, sampler, n_samples)
    gps = _tracegeodesics(m, xs, vs, g, λmax, callback=callback, **kwargs)
    mask = gps["status"] == StatusCodes.IntersectedWithGeometry
    return CoronaGeodesics(m, g, model, gps[mask], vsrc[mask])


# ------------------------------------------------------------------------------------------
# disc kinematics on arrays of points
# ------------------------------------------------------------------------------------------
def _components(m, r, θ):
    return m._components(r, np.sin(θ), np.cos(θ))


def _equatorial_project(x):
    return x[..., 1] * np.abs(np.sin(x[..., 2]))


def circular_fourvelocity(m, r):
    """CircularOrbits.fourvelocity(m, r) at θ = π/2 for an array of radii (circular-orbits.jl:11-37,114-123)."""
    from .special_radii import Jet

    r = np.asarray(r, dtype=np.float64)
    g = [Jet.lift(c) for c in m._components(Jet(r, np.ones_like(r), np.zeros_like(r)), 1.0, 0.0)]
    gv = [c.v + 0.0 * r for c in g]
    dg = [c.d + 0.0 * r for c in g]
    Om = -(dg[4] - np.sqrt(dg[4] * dg[4] - dg[0] * dg[3])) / dg[3]
    D = gv[0] * gv[3] - gv[4] * gv[4]
    itt, ipp, itp = gv[3] / D, gv[0] / D, -gv[4] / D
    A = -(Om * itt - itp)
    B = Om * itp - ipp
    den = B * B * itt + 2.0 * A * B * itp + A * A * ipp
    d = -np.sign(den) * np.sqrt(1.0 / np.abs(den))
    ut, up = B * d, A * d
    out = np.zeros(r.shape + (4,))
    out[..., 0] = itt * ut + itp * up
    out[..., 3] = itp * ut + ipp * up
    return out


def _nan_linear_interp(t, u, x, default=0.0):
    """NaNLinearInterpolator (interpolations.jl:1-30): linear, extrapolating from the end intervals."""
    t, u, x = np.asarray(t), np.asarray(u), np.asarray(x, dtype=np.float64)
    idx = np.clip(np.searchsorted(t, x, side="right"), 1, t.size - 1) - 1
    x1, x2, y1, y2 = t[idx], t[idx + 1], u[idx], u[idx + 1]
    w = (x - x1) / (x2 - x1)
    y = (1.0 - w) * y1 + w * y2
    bad = np.isnan(y)
    if np.any(bad):
        pick = np.where(w < 0.5, y1, y2)
        y = np.where(bad, np.where(np.isnan(pick), default, pick), y)
    return y


def keplerian_velocity_projector(m, plunging=None, ensemble=None):
    """_keplerian_velocity_projector(m) (circular-orbits.jl:155-170): x (n, 4) -> disc four-velocity,
    Keplerian outside the ISCO, the tabulated plunge (v^r sign flipped) inside.  `plunging` is the
    (r, v^t, v^r, v^ϕ) table; by default it is traced on the device (interpolate_plunging_velocities)."""
    r_isco = m.isco()
    table = [plunging]

    def project(x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            return project(x[None, :])[0]
        r = _equatorial_project(x)
        out = circular_fourvelocity(m, np.where(r < r_isco, r_isco, r))
        inside = r < r_isco
        if np.any(inside):
            if table[0] is None:
                from .special_radii import interpolate_plunging_velocities

                table[0] = interpolate_plunging_velocities(m, ensemble=ensemble)
            tr, tvt, tvr, tvp = table[0]
            ri = r[inside]
            out[inside, 0] = _nan_linear_interp(tr, tvt, ri)
            out[inside, 1] = -_nan_linear_interp(tr, tvr, ri)
            out[inside, 2] = 0.0
            out[inside, 3] = _nan_linear_interp(tr, tvp, ri)
        return out

    return project


def energy_ratio(m, gps, v_src, v_disc):
    """flux-calculations.jl:96-110: e_src / e_disc for arrays of GeodesicPoint records."""
    gs = _components(m, gps["x_init"][:, 1], gps["x_init"][:, 2])
    gd = _components(m, gps["x"][:, 1], gps["x"][:, 2])
    return _dot(gs, gps["v_init"], v_src) / _dot(gd, gps["v"], v_disc)


def lorentz_factor(m, x, v):
    """lorentz_factor(m, x, v; component = 4) (flux-calculations.jl:20-40): 𝒱^ϕ measured in the
    LNRF, whose ϕ and t legs are √g_ϕϕ (dϕ - ω dt) and √(-1/g^tt) dt for every static axis-symmetric
    metric (the closed form of `lnrbasis`; checked against the Gram-Schmidt basis in the tests)."""
    x, v = np.asarray(x, dtype=np.float64), np.asarray(v, dtype=np.float64)
    g = _components(m, x[..., 1], x[..., 2])
    om = -g[4] / g[3]
    alpha = np.sqrt(-g[0] + g[4] * g[4] / g[3])
    V = np.sqrt(g[3]) * (v[..., 3] - om * v[..., 0]) / (alpha * v[..., 0])
    return 1.0 / np.sqrt(1.0 - V * V)


def _proper_area(m, r, θ):
    g = _components(m, np.asarray(r, dtype=np.float64), np.asarray(θ, dtype=np.float64))
    return TWO_PI * np.sqrt(g[1] * g[3])


@dataclass(frozen=True)
class PowerLawSpectrum:
    Γ: float = 2.0


def coronal_spectrum(spectrum, g):
    return g ** (-spectrum.Γ)


def point_source_equatorial_disc_emissivity(spec, θ, g, A, γ):
    return np.abs(np.sin(θ)) * coronal_spectrum(spec, g) / (A * γ)


def source_to_disc_emissivity(m, spec, N, A, x, g, v_disc):
    return N * coronal_spectrum(spec, g) / (A * lorentz_factor(m, x, v_disc))


# ------------------------------------------------------------------------------------------
# radial profiles (src/corona/radial.jl)
# ------------------------------------------------------------------------------------------
@dataclass
class RadialDiscProfile:
    radii: np.ndarray
    ε: np.ndarray
    t: np.ndarray
    _warned: list = field(default_factory=list, repr=False)

    def _bounded(self, r):
        return np.clip(r, self.radii[0], self.radii[-1])

    def emissivity_at(self, r):
        return _nan_linear_interp(self.radii, self.ε, self._bounded(np.asarray(r, dtype=np.float64)))

    def coordtime_at(self, r):
        return _nan_linear_interp(self.radii, self.t, self._bounded(np.asarray(r, dtype=np.float64)))


def emissivity_at(prof, r):
    return prof.emissivity_at(r)


def coordtime_at(prof, r):
    return prof.coordtime_at(r)


def _bucket_index(values, bins):
    """Buckets.Simple(): the bin of a value is the last edge <= value (clamped to the ends).  This is
    what the reference's golden emissivity vector implies: with "first edge >= value" the first bin
    would hold one photon and a 14x larger emissivity than recorded."""
    return np.clip(np.searchsorted(bins, values, side="right") - 1, 0, bins.size - 1)


def _profile_from_bins(m, spec, bins, count, gsum, tsum, grouped, disc_velocity):
    """the second half of _build_radial_profile (radial.jl:70-100): per-bin means -> emissivity."""
    with np.errstate(invalid="ignore", divide="ignore"):
        gs = gsum / count          # mean per bin; NaN when empty
        ts = tsum / count
    g_at = _nan_linear_interp(bins, gs, bins)
    dr = np.diff(np.concatenate([[0.0], bins]))
    xb = np.zeros((bins.size, 4))
    xb[:, 1], xb[:, 2] = bins, math.pi / 2
    vb = disc_velocity(xb)
    A = dr * _proper_area(m, bins, math.pi / 2)
    with np.errstate(all="ignore"):          # empty bins: 0 photons x (0 redshift)^-Γ = NaN, as in the reference
        ε = source_to_disc_emissivity(m, spec, grouped, A, xb, g_at, vb)
    return RadialDiscProfile(bins, ε, ts)


def build_radial_profile(m, spec, points, source_velocities, *, grid=None, N=100, intensity=None, disc_velocity=None,
                         ensemble=None):
    """_build_radial_profile + the sorting wrapper (radial.jl:38-100,132-141,155-165)."""
    grid = GeometricGrid() if grid is None else grid
    disc_velocity = keplerian_velocity_projector(m, ensemble=ensemble) if disc_velocity is None else disc_velocity
    radii = _equatorial_project(points["x"])
    J = np.argsort(radii, kind="stable")
    points, source_velocities, radii = points[J], np.asarray(source_velocities)[J], radii[J]
    times = points["x"][:, 0]
    bins = np.asarray(grid(radii.min(), radii.max(), N), dtype=np.float64)
    idx = _bucket_index(radii, bins)
    count = np.bincount(idx, minlength=bins.size).astype(np.float64)
    g_all = energy_ratio(m, points, source_velocities, disc_velocity(points["x"]))
    gsum = np.bincount(idx, weights=g_all, minlength=bins.size)
    tsum = np.bincount(idx, weights=times, minlength=bins.size)
    grouped = count if intensity is None else np.bincount(idx, weights=np.asarray(intensity)[J], minlength=bins.size)
    return _profile_from_bins(m, spec, bins, count, gsum, tsum, grouped, disc_velocity)


# ------------------------------------------------------------------------------------------
# corona -> disc on the device (gr_corona_trace / gr_corona_bin)
# ------------------------------------------------------------------------------------------
def _sky_rows_batch(m, model, n, rmin):
    """The 28-double rows of n samples of a source without one position, all at once (a Python loop over the samples costs 0.3 ms
    each: five minutes for 10⁶): positions from the model's own generator in the order its sample_position_velocity would draw them,
    velocities from SourceVelocities.batch, tetrads from tetradframe_batch.  None when the model's velocity function is not one this
    module can evaluate on arrays (the caller then loops)."""
    state = model.rng.bit_generator.state
    xs = model.sample_positions(m, n, rmin)
    with np.errstate(invalid="ignore"):
        vs = SourceVelocities.batch(model.vf, m, xs)          # (the velocity belongs to the position as drawn, the clamp comes after)
    xs[:, 2] = np.clip(xs[:, 2], 1e-3, math.pi - 1e-3)
    if vs is None or not np.all(np.isfinite(vs)):
        model.rng.bit_generator.state = state      # nothing drawn
        return None
    s, c = np.sin(xs[:, 2]), np.cos(xs[:, 2])
    g = [np.asarray(q, dtype=np.float64) + 0.0 * s for q in m._components(xs[:, 1], s, c)]
    G4 = np.zeros((n, 4, 4))
    G4[:, 0, 0], G4[:, 1, 1], G4[:, 2, 2], G4[:, 3, 3] = g[0], g[1], g[2], g[3]
    G4[:, 0, 3] = G4[:, 3, 0] = g[4]
    T = np.stack(tetradframe_batch(G4, vs), axis=2)            # columns: the four vectors (tetradframe_matrix)
    sp, cp = np.sin(xs[:, 3]), np.cos(xs[:, 3])
    B = np.zeros((n, 4, 4))
    B[:, 0, 0] = 1.0
    B[:, 1, 1], B[:, 1, 2], B[:, 1, 3] = s * cp, s * sp, c                    # _cart_to_spher_jacobian
    B[:, 2, 1], B[:, 2, 2], B[:, 2, 3] = c * cp, c * sp, -s
    B[:, 3, 1], B[:, 3, 2] = -sp, cp
    rows = np.zeros((n, 28))
    rows[:, 0:4] = xs
    rows[:, 4:20] = np.einsum("nij,njk->nik", T, B).reshape(n, 16)
    rows[:, 20] = g[0] * vs[:, 0] + g[4] * vs[:, 3]
    rows[:, 21], rows[:, 22] = g[1] * vs[:, 1], g[2] * vs[:, 2]
    rows[:, 23] = g[3] * vs[:, 3] + g[4] * vs[:, 0]
    rows[:, 24], rows[:, 27] = g[0], g[4]
    return rows, vs


def sky_rayset(m, model, sampler, n_samples):
    """The gr_rayset of `tracegeodesics(m, model, ...)` for a source at one position: the tetrad and the
    Cartesian -> spherical Jacobian go in as one matrix (v = T (1, J k̂), samplers.jl:81-99) and sample number ->
    sky angles -> k̂ happens per lane on the device (samplers.jl:30-44).  Returns (rayset, keepalive, x, v_source)."""
    from . import _lib

    rmin = m.inner_radius() * 1.9

    def matrix_at(x, v):
        B = np.eye(4)
        B[1:, 1:] = _cart_to_spher_jacobian(x[2], x[3])
        return tetradframe_matrix(m, x, v) @ B

    rs = _lib.gr_rayset()
    rows = None
    if model.point_source or model.fixed_position:
        x, v = model.sample_position_velocity(m)
        if x[1] < rmin:
            raise ValueError("source position lies inside 1.9 inner radii")
        x = np.array(x, dtype=np.float64)
        x[2] = min(max(x[2], 1e-3), math.pi - 1e-3)             # avoid coordinate singularities, corona-models.jl:18-24
        v = np.asarray(v, dtype=np.float64)
        Mx = matrix_at(x, v)
    else:
        # A source without one position (DiscCorona; any model whose sample_position_velocity draws): every sample brings its own
        # position, matrix and -- for the energy ratio -- its source velocity with the index lowered, and g_tμ there: 28 doubles
        # (gr_rayset.sky_rows), in the order corona-models.jl:1-33 draws them (rejecting positions inside 1.9 inner radii).
        rows = np.zeros((int(n_samples), 28))
        vsrc = np.zeros((int(n_samples), 4))
        batch = _sky_rows_batch(m, model, int(n_samples), rmin) if hasattr(model, "sample_positions") else None
        if batch is not None:
            rows, vsrc = batch
        for k in range(int(n_samples) if batch is None else 0):
            xk, vk = model.sample_position_velocity(m)
            redraws = 0
            while xk[1] < rmin:
                xk, vk = model.sample_position_velocity(m)
                redraws += 1
                if redraws > 100_000:
                    raise ValueError("source positions lie inside 1.9 inner radii")
            xk = np.array(xk, dtype=np.float64)
            xk[2] = min(max(xk[2], 1e-3), math.pi - 1e-3)
            vk = np.asarray(vk, dtype=np.float64)
            g = _components_at(m, xk)
            rows[k, 0:4] = xk
            rows[k, 4:20] = matrix_at(xk, vk).ravel()
            rows[k, 20:24] = (g[0] * vk[0] + g[4] * vk[3], g[1] * vk[1], g[2] * vk[2], g[3] * vk[3] + g[4] * vk[0])
            rows[k, 24:28] = (g[0], 0.0, 0.0, g[4])
            vsrc[k] = vk
        x, v, Mx = rows[int(np.argmax(rows[:, 1])), 0:4].copy(), vsrc, np.eye(4)      # (x: the outermost sample, for the chart's range checks)
        rs.sky_rows = rows.ctypes.data
    for q in range(4):
        rs.x_obs[q] = x[q]
    for q, val in enumerate(np.ascontiguousarray(Mx).ravel()):
        rs.Mx[q] = val
    rs.n = int(n_samples)
    if isinstance(sampler, EvenSampler):
        rs.sky_sampler = 1
    elif isinstance(sampler, WeierstrassSampler):
        rs.sky_sampler, rs.sky_resolution = 2, sampler.resolution
    else:
        raise NotImplementedError(f"Not implemented for {type(sampler).__name__}.")
    rs.sky_both = 1 if isinstance(sampler.domain, BothHemispheres) else 0
    keep = None
    g = sampler.generator
    if isinstance(g, GoldenSpiralGenerator):
        rs.sky_generator = 0
    elif isinstance(g, EvenGenerator):
        rs.sky_generator = 1
    else:
        # any other generator (RandomGenerator: rand() N): its numbers cross, 8 bytes per ray
        keep = np.ascontiguousarray(geti(sampler, np.arange(1, n_samples + 1), n_samples), dtype=np.float64)
        rs.sky_generator, rs.sky_i = 2, keep.ctypes.data
    return rs, (None if keep is None and rows is None else (keep, rows)), x, v


_PLUNGING_TABLES = {}


def _plunging_table(m, ensemble):
    """interpolate_plunging_velocities(m) once per metric value: one saved geodesic, a few ms, reused by every profile."""
    from .special_radii import interpolate_plunging_velocities

    try:
        key = (type(m).__name__, int(m.metric_id)) + tuple(float(p) for p in m.abi_params())
        if m.metric_id == 11:
            key += (float(m.table[8]),)          # the table's build id (unique per fit; an id() is reused once its object is gone)
    except Exception:          # a metric without flat parameters: no reuse
        return tuple(interpolate_plunging_velocities(m, ensemble=ensemble))
    if key not in _PLUNGING_TABLES:
        if len(_PLUNGING_TABLES) >= 16:
            _PLUNGING_TABLES.pop(next(iter(_PLUNGING_TABLES)))
        _PLUNGING_TABLES[key] = tuple(interpolate_plunging_velocities(m, ensemble=ensemble))
    return _PLUNGING_TABLES[key]


def device_radial_profile(m, d, model, spectrum=None, *, λmax=10_000.0, sampler=None, n_samples=1000, grid=None, N=100,
                          callback="default", ensemble=None, stats=False, **solver_args):
    """emissivity_profile(m, d, model, spectrum; n_samples, sampler, N, grid) (emissivity.jl:118-168) with the whole
    per-ray half on the device: sky sampling, tracing, energy_ratio against the Keplerian / plunging disc velocity
    (gr_corona_trace) and the radial bucketing with the per-bin sums of g and t (gr_corona_bin).  What comes back
    over PCIe is (ρ_min, ρ_max, hits) and 3 N doubles; the N-bin tail of radial.jl:86-100 runs on the host."""
    import ctypes as C

    from . import _lib
    from .rendering import abi_pointfunction
    from .tracing import tracing_configuration

    spectrum = PowerLawSpectrum(2.0) if spectrum is None else spectrum
    sampler = EvenSampler(BothHemispheres(), GoldenSpiralGenerator()) if sampler is None else sampler
    grid = GeometricGrid() if grid is None else grid
    if callback == "default":
        callback = domain_upper_hemisphere()
    rs, keep, x, v_src = sky_rayset(m, model, sampler, n_samples)
    config = tracing_configuration(m, x, np.zeros((1, 4)), d, (0.0, λmax), callback=callback, ensemble=ensemble, **solver_args)
    ens = config.ensemble
    # the disc velocity of _keplerian_velocity_projector (circular-orbits.jl:155-170): ALWAYS the traced plunging
    # table inside the ISCO (the analytic Kerr plunge of the image-plane redshift is a different, if close, function)
    plunging = _plunging_table(m, ens)
    disc_velocity = keplerian_velocity_projector(m, plunging=plunging, ensemble=ens)
    from .pointfunctions import GR_PF_REDSHIFT, PointFunction

    rpf = PointFunction(None, device_pf=GR_PF_REDSHIFT, extra={"r_isco": m.isco(), "plunge": plunging})
    pf, keep_pf = abi_pointfunction(rpf)
    if np.ndim(v_src) == 1:
        pf.has_u_src = 1
        for q in range(4):
            pf.u_src[q] = v_src[q]
    # (a source of many positions: the rows of the ray set carry each sample's source velocity, pf.has_u_src stays 0)
    cfg = config.abi_config()
    L = _lib.load()
    lim = np.zeros(2)
    hits = C.c_int64(0)
    st = _lib.gr_stats() if stats else None
    many = ens.multi
    if many:
        # several devices: the samples shard like any ray set (gr_rayset.sky_first / sky_total, ABI 8); every context keeps and bins
        # its own rows, the integer accumulators add up to the bits one context would give
        ctx_arr, ctx_stats = _lib.ctx_array(ens.contexts)
        _lib.check(L.gr_corona_trace_multi(ctx_arr, len(ens.contexts), C.byref(cfg), C.byref(rs), C.byref(pf), lim.ctypes.data, C.byref(hits),
                                           ctx_stats))
        if stats:
            st = _lib.merge_stats(ctx_stats)
    else:
        _lib.check(L.gr_corona_trace(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), lim.ctypes.data, C.byref(hits),
                                     C.byref(st) if stats else None))
    if hits.value == 0:
        raise ValueError("no ray of the corona reached the disc")
    bins = np.ascontiguousarray(grid(lim[0], lim[1], N), dtype=np.float64)
    out = np.zeros((3, bins.size))
    if many:
        _lib.check(L.gr_corona_bin_multi(ctx_arr, len(ens.contexts), bins.ctypes.data, bins.size, out.ctypes.data))
    else:
        _lib.check(L.gr_corona_bin(ens.ctx.handle, bins.ctypes.data, bins.size, out.ctypes.data))
    prof = _profile_from_bins(m, spectrum, bins, out[0], out[1], out[2], out[0], disc_velocity)
    return (prof, st) if stats else prof


def _point_source_emissivity(m, spec, source_velocity, r, δs, points, disc_velocity):
    """lamp-post.jl:118-156 on sorted arrays."""
    v_disc = disc_velocity(points["x"])
    n = r.size
    vsrc = np.tile(np.asarray(source_velocity, dtype=np.float64), (n, 1))
    gs = energy_ratio(m, points, vsrc, v_disc)
    γ = lorentz_factor(m, points["x"], v_disc)
    k = np.arange(n)
    i2 = np.where(k == 0, 1, np.where(k != n - 1, k + 1, k - 1))
    i4 = np.where(k == 0, 1, k - 1)
    Δr = (np.abs(r[k] - r[i2]) + np.abs(r[k] - r[i4])) / 2
    weight = (np.abs(δs[k] - δs[i2]) + np.abs(δs[k] - δs[i4])) / 4
    A = _proper_area(m, points["x"][:, 1], points["x"][:, 2]) * Δr
    return weight * point_source_equatorial_disc_emissivity(spec, δs, gs, A, γ), gs


def point_source_profile_from_points(m, spec, source_velocity, δs, gps, disc_velocity):
    """the reduction half of _point_source_symmetric_emissivity_profile (lamp-post.jl:97-115)"""
    I = gps["status"] == StatusCodes.IntersectedWithGeometry
    points, δs = gps[I], np.asarray(δs)[I]
    rs = _equatorial_project(points["x"])
    J = np.argsort(rs, kind="stable")
    rs, points, δs = rs[J], points[J], δs[J]
    ε, _ = _point_source_emissivity(m, spec, source_velocity, rs, δs, points, disc_velocity)
    return RadialDiscProfile(rs, ε, points["x"][:, 0].copy())


def polar_angle_velocities(m, x, v, δs, ϕ=0.0):
    """polar_angle_to_velfunc (emissivity.jl:175-179) evaluated for every δ."""
    return sky_angles_to_velocity(m, x, v, np.asarray(δs, dtype=np.float64), np.full(len(δs), ϕ))


def emissivity_profile(m, d, model, spectrum=None, *, λmax=10_000.0, δmin=0.01, δmax=179.99, sampler=None,
                       n_samples=1000, grid=None, N=100, callback="default", ensemble=None, **kwargs):
    """emissivity_profile(m, d, model, [spectrum]; n_samples, sampler, N, grid) emissivity.jl:118-168;
    point sources without an explicit sampler take the angular method of lamp-post.jl:68-115,158-166."""
    spectrum = PowerLawSpectrum(2.0) if spectrum is None else spectrum
    if callback == "default":
        callback = domain_upper_hemisphere()
    disc_velocity = keplerian_velocity_projector(m, ensemble=ensemble)
    if sampler is None and model.point_source:
        δs = np.radians(np.linspace(δmin, δmax, n_samples))
        x, v = model.sample_position_velocity(m)
        vs = polar_angle_velocities(m, x, v, δs)
        gps = _tracegeodesics(m, x, vs, d, λmax, callback=callback, ensemble=ensemble, **kwargs)
        return point_source_profile_from_points(m, spectrum, v, δs, gps, disc_velocity)
    sampler = EvenSampler(BothHemispheres(), GoldenSpiralGenerator()) if sampler is None else sampler
    if _sky_route(m, model, sampler):
        return device_radial_profile(m, d, model, spectrum, λmax=λmax, sampler=sampler, n_samples=n_samples, grid=grid, N=N,
                                     callback=callback, ensemble=ensemble, **kwargs)
    cg = tracecorona(m, d, model, sampler=sampler, λmax=λmax, n_samples=n_samples, ensemble=ensemble)
    return build_radial_profile(m, spectrum, cg.geodesic_points, cg.source_velocity, grid=grid, N=N,
                                disc_velocity=disc_velocity)
