"""Host-side mirror of the reference's AbstractStaticAxisSymmetric metrics.

Only what the render path needs on the host: parameters -> (metric_id, params[8]) for the C
ABI, `metric_components` for the one-off observer set-up (LNRF basis), `inner_radius` and
`isco`.  The per-ray evaluation (with derivatives) happens in the HIP kernels.

Reference: src/metrics/kerr-metric.jl:11-28,62-72,91; src/metrics/johannsen-ad.jl:4-34,49-67;
src/metrics/kerr-metric-first-order.jl:297-337 (Z1, Z2, isco).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass

import numpy as np

GR_METRIC_KERR, GR_METRIC_JOHANNSEN = 0, 1
GR_METRIC_MORRIS_THORNE, GR_METRIC_BUMBLEBEE, GR_METRIC_KERR_NEWMAN, GR_METRIC_JOHANNSEN_PSALTIS = 2, 3, 4, 5
GR_METRIC_DILATON_AXION = 6
GR_METRIC_SPHERICAL, GR_METRIC_KERR_DARK_MATTER, GR_METRIC_KERR_REFRACTIVE, GR_METRIC_NOZ = 7, 8, 9, 10
GR_METRIC_TABULATED = 11


class AbstractMetric:
    metric_id: int = -1

    def abi_params(self):
        raise NotImplementedError

    def metric_components(self, r, theta):
        """(g_tt, g_rr, g_θθ, g_ϕϕ, g_tϕ) at (r, θ)."""
        raise NotImplementedError(f"Not implemented for {type(self).__name__}.")

    def inner_radius(self):
        raise NotImplementedError(f"Not implemented for {type(self).__name__}.")

    # metric(m, x) = _symmetric_matrix(comps); auto-diff.jl:228-232, utils.jl:60-67
    def metric(self, x):
        r, th = (x[1], x[2]) if len(x) == 4 else (x[0], x[1])
        g = self.metric_components(r, th)
        G = np.zeros((4, 4))
        G[0, 0], G[1, 1], G[2, 2], G[3, 3] = g[0], g[1], g[2], g[3]
        G[0, 3] = G[3, 0] = g[4]
        return G


class AbstractStaticAxisSymmetric(AbstractMetric):
    pass


@dataclass(frozen=True)
class KerrMetric(AbstractStaticAxisSymmetric):
    """KerrMetric(M = 1.0, a = 0.0) -- src/metrics/kerr-metric.jl:62-69."""

    M: float = 1.0
    a: float = 0.0
    metric_id = GR_METRIC_KERR

    def abi_params(self):
        return [self.M, self.a]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        """kerr-metric.jl:11-28; `r` may be a float or a special_radii.Jet."""
        M, a = self.M, self.a
        R = 2.0 * M
        s2 = s * s
        c2 = 1.0 - s2
        Sig = r * r + a * a * c2
        iSig = 1.0 / Sig
        gam = s2 * R * r * a
        tt = -(1.0 - (R * r) * iSig)
        rr = Sig / (r * r + a * a - R * r)
        pp = s2 * (r * r + a * a + (gam * a) * iSig)
        tp = -gam * iSig
        return (tt, rr, Sig, pp, tp)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        return kerr_isco(self.M, self.a)


@dataclass(frozen=True)
class JohannsenMetric(AbstractStaticAxisSymmetric):
    """JohannsenMetric(M, a, α13, α22, α52, ϵ3) -- src/metrics/johannsen-ad.jl:49-63."""

    M: float = 1.0
    a: float = 0.0
    alpha13: float = 0.0
    alpha22: float = 0.0
    alpha52: float = 0.0
    eps3: float = 0.0
    metric_id = GR_METRIC_JOHANNSEN

    def abi_params(self):
        return [self.M, self.a, self.alpha13, self.alpha22, self.alpha52, self.eps3]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        """johannsen-ad.jl:4-34; `r` may be a float or a special_radii.Jet."""
        M, a = self.M, self.a
        Mr = M / r
        A1 = 1.0 + self.alpha13 * (Mr * Mr * Mr)
        A2 = 1.0 + self.alpha22 * (Mr * Mr)
        A5 = 1.0 + self.alpha52 * (Mr * Mr)
        Sig = r * r + a * a * (c * c) + (self.eps3 * M ** 3) / r
        Del = r * r - 2.0 * M * r + a * a
        r2a2 = r * r + a * a
        s2 = s * s
        dn = r2a2 * A1 - (a * a * s2) * A2
        denom = dn * dn
        tt = -(Sig * (Del - (a * a * s2) * (A2 * A2)))
        rr = Sig / (Del * A5)
        pp = (Sig * s2) * ((r2a2 * r2a2) * (A1 * A1) - (a * a * s2) * Del)
        tp = -(a * ((Sig * s2) * (r2a2 * A1 * A2 - Del)))
        return (tt / denom, rr, Sig, pp / denom, tp / denom)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)


@dataclass(frozen=True)
class MorrisThorneWormhole(AbstractStaticAxisSymmetric):
    """MorrisThorneWormhole(b = 1.0) -- src/metrics/morris-thorne-ad.jl:4-39."""

    b: float = 1.0
    metric_id = GR_METRIC_MORRIS_THORNE

    def abi_params(self):
        return [self.b]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, l, s, c):
        w = l * l + self.b ** 2
        # ϕϕ = (b² + l²) sin(θ): first power of sinθ, as in the reference
        return (-1.0 + 0.0 * l, 1.0 + 0.0 * l, w, w * s, 0.0 * l)

    def inner_radius(self):
        return 0.0


@dataclass(frozen=True)
class BumblebeeMetric(AbstractStaticAxisSymmetric):
    """BumblebeeMetric(M, a, l) -- src/metrics/bumblebee-ad.jl:6-52 (slow rotation, |a| < 0.3)."""

    M: float = 1.0
    a: float = 0.0
    l: float = 0.0
    metric_id = GR_METRIC_BUMBLEBEE

    def __post_init__(self):
        if self.l <= -1.0:
            raise ValueError("l must be >-1")
        if abs(self.a) > 0.3:
            raise ValueError("This metric is for the slow rotation approximation only, and requires |a| < 0.3.")

    def abi_params(self):
        return [self.M, self.a, self.l]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        M, a, l = self.M, self.a, self.l
        s2 = s * s
        Del = (r * r - 2.0 * M * r) / (l + 1.0)
        return (-(1.0 - 2.0 * M / r), r * r / Del, r * r, r * r * s2, (-2.0 * M * a * s2) / r)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)


@dataclass(frozen=True)
class KerrNewmanMetric(AbstractStaticAxisSymmetric):
    """KerrNewmanMetric(M, a, Q) -- src/metrics/kerr-newman-ad.jl:6-64.  Null / uncharged
    geodesics only: the Lorentz-force term of a charged test particle (q ≠ 0) is not on the device."""

    M: float = 1.0
    a: float = 0.0
    Q: float = 0.0
    metric_id = GR_METRIC_KERR_NEWMAN

    def __post_init__(self):
        if self.a ** 2 + self.Q ** 2 > self.M ** 2:
            raise ValueError("Value error: `a^2 + Q^2` must be `<= M^2`")

    def abi_params(self):
        return [self.M, self.a, self.Q]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        M, a, Q = self.M, self.a, self.Q
        Sig = r * r + (a * c) ** 2
        s2 = s * s
        Del = r * r - 2.0 * M * r + a * a + Q * Q
        r2a2 = r * r + a * a
        return ((a * a * s2 - Del) / Sig, Sig / Del, Sig, (s2 / Sig) * (r2a2 * r2a2 - a * a * s2 * Del),
                (a * s2 / Sig) * (Del - r2a2))

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2 - self.Q ** 2)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)


@dataclass(frozen=True)
class JohannsenPsaltisMetric(AbstractStaticAxisSymmetric):
    """JohannsenPsaltisMetric(M, a, ϵ3) -- src/metrics/johannsen-psaltis-ad.jl:4-46."""

    M: float = 1.0
    a: float = 0.0
    eps3: float = 0.0
    metric_id = GR_METRIC_JOHANNSEN_PSALTIS

    def abi_params(self):
        return [self.M, self.a, self.eps3]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        M, a, e3 = self.M, self.a, self.eps3
        Sig = r * r + a * a * (c * c)
        h = (e3 * M ** 3) * r / (Sig * Sig)
        s2 = s * s
        Del = r * r - 2.0 * M * r + a * a
        tMr = 2.0 * M * r
        tt = -((1.0 + h) * (1.0 - tMr / Sig))
        rr = Sig * (1.0 + h) / (Del + (a * a * s2) * h)
        term1 = s2 * (r * r + a * a + (a * a * s2) * tMr / Sig)
        term2 = (h * (a * a)) * (Sig + tMr) * (s2 * s2) / Sig
        tp = -((a * tMr) * (s2 * (1.0 + h)) / Sig)
        return (tt, rr, Sig, term1 + term2, tp)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)


@dataclass(frozen=True)
class DilatonAxion(AbstractStaticAxisSymmetric):
    """DilatonAxion(M, a, β, b): Einstein-Maxwell-dilaton-axion metric -- src/metrics/dilaton-axion-ad.jl:8-75."""

    M: float = 1.0
    a: float = 0.5
    β: float = 0.0
    b: float = 1.0
    metric_id = GR_METRIC_DILATON_AXION

    def abi_params(self):
        return [self.M, self.a, self.β, self.b]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        M, a, be, b = self.M, self.a, self.β, self.b
        R = M
        bb = 0.0 if be == 0.0 else be / b
        ba = 0.0 if be == 0.0 else be / a
        bab = 0.0 if be == 0.0 else be / (a * b)
        s2 = s * s
        Sig = r * r + a * a * (c * c)
        Del = r * r + a * a - (2.0 * R) * r
        bt = (2.0 * b) * r + be * be
        Delh = Del - bt - R * (R + 2.0 * b) * bb * bb
        Sigh = Sig - bt + (R * R * bb) * (bb - 2.0 * a * c)
        de = r * r - (2.0 * b) * r + a * a
        W = 1.0 + (bab * (2.0 * c - bab) + ba * ba) / s2
        Was = W * a * s
        A = de * de - Delh * (Was * Was)
        return (-((Delh - a * a * s2) / Sigh), Sigh / Delh, Sigh, (A * s2) / Sigh, -((a * (de - Delh * W)) * s2 / Sigh))

    def inner_radius(self):
        M, a, be, b = self.M, self.a, self.β, self.b
        bb = 0.0 if be == 0.0 else be / b
        return M + b + math.sqrt((M + b) ** 2 - a * a + be * be - (M - 2.0 * b) * M * bb * bb)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)


def _where(cond, a, b):
    """a where cond else b, for floats, arrays and Jets of either (piecewise metric functions)."""
    from .special_radii import Jet

    if isinstance(a, Jet) or isinstance(b, Jet):
        a, b = Jet.lift(a), Jet.lift(b)
        return Jet(np.where(cond, a.v, b.v) + 0.0, np.where(cond, a.d, b.d) + 0.0, np.where(cond, a.dd, b.dd) + 0.0)
    out = np.where(cond, a, b)
    return float(out) if np.ndim(out) == 0 else out


def _value(x):
    return x.v if hasattr(x, "dd") else x


@dataclass(frozen=True)
class SphericalMetric(AbstractStaticAxisSymmetric):
    """SphericalMetric(): flat space in spherical coordinates -- src/metrics/minkowski.jl:1-15."""

    metric_id = GR_METRIC_SPHERICAL

    def abi_params(self):
        return []

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        return (-1.0 + 0.0 * r, 1.0 + 0.0 * r, r * r, r * r * (s * s), 0.0 * r)

    def inner_radius(self):
        return 4.0 * float(np.finfo(np.float64).eps)

    def isco(self):
        return 0.0


@dataclass(frozen=True)
class KerrDarkMatter(AbstractStaticAxisSymmetric):
    """KerrDarkMatter(M, a, M_dark_matter, Δr, rₛ) -- src/metrics/kerr-dark-matter.jl:6-70 (arXiv:2003.06829):
    the Kerr metric with the mass M + M_dark_matter G((r - rₛ)/Δr) enclosed at radius r."""

    M: float = 1.0
    a: float = 0.0
    M_dark_matter: float = 2.0
    Δr: float = 20.0
    rₛ: float = 10.0
    metric_id = GR_METRIC_KERR_DARK_MATTER

    def abi_params(self):
        return [self.M, self.a, self.M_dark_matter, self.Δr, self.rₛ]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        a = self.a
        rv = _value(r)
        dr = (r - self.rₛ) / self.Δr
        G = (3.0 - 2.0 * dr) * (dr * dr)
        Mdm = _where(rv < self.rₛ, 0.0 * r, _where(rv < self.rₛ + self.Δr, self.M_dark_matter * G, self.M_dark_matter + 0.0 * r))
        R = 2.0 * (self.M + Mdm)
        s2 = s * s
        Sig = r * r + a * a * (1.0 - s2)
        Rr = R * r
        return (-(1.0 - Rr / Sig), Sig / (r * r + a * a - Rr), Sig, s2 * (r * r + a * a + (a * a * s2) * Rr / Sig),
                -(a * Rr * s2) / Sig)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)

    def break_radii(self):
        """Where `metric_components` changes form (TabulatedMetric `breaks`): the enclosed mass is piecewise, C¹ at rₛ and rₛ + Δr
        (kerr-dark-matter.jl:12-20)."""
        return [(self.rₛ, 0.0), (self.rₛ + self.Δr, 0.0)]


def _smooth_interpolate(x, x0, δx=2.5, smoothing_offset=1e4):
    """utils.jl:158-168 (floats or arrays)"""
    if np.ndim(x):
        x = np.asarray(x, dtype=np.float64)
        mid = 1.0 - (np.arctan(smoothing_offset * (x - x0) / δx) / math.pi + 0.5)
        return np.where(x <= x0 - δx / 2, 1.0, np.where(x <= x0 + δx / 2, mid, 0.0))
    if x <= x0 - δx / 2:
        return 1.0
    if x <= x0 + δx / 2:
        return 1.0 - (math.atan(smoothing_offset * (x - x0) / δx) / math.pi + 0.5)
    return 0.0


@dataclass(frozen=True)
class KerrRefractive(AbstractStaticAxisSymmetric):
    """KerrRefractive(M, a, n, corona_radius) -- src/metrics/kerr-refractive-ad.jl:8-64: Kerr with a
    path-length ansatz equivalent to a refractive index n inside the corona radius."""

    M: float = 1.0
    a: float = 0.0
    n: float = 1.0
    corona_radius: float = 20.0
    metric_id = GR_METRIC_KERR_REFRACTIVE

    def abi_params(self):
        return [self.M, self.a, self.n, self.corona_radius]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        a = self.a
        R = 2.0 * self.M
        Sig = r * r + a * a * (c * c)
        s2 = s * s
        rv = _value(r)
        t = _smooth_interpolate(rv if np.ndim(rv) else float(rv), self.corona_radius)
        n = t + (1.0 - t) * self.n
        return (-(1.0 - R * r / Sig) / (n * n), Sig / (r * r - R * r + a * a), Sig,
                s2 * (r * r + a * a + (s2 * R * r * a * a) / Sig), (-(R * r * a * s2) / Sig) / n)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        return kerr_isco(self.M, self.a)                 # kerr-refractive-ad.jl:61

    def break_radii(self):
        """Where `metric_components` changes form (TabulatedMetric `breaks`): the refractive index jumps by 6e-5 at
        corona_radius ± δx/2 and steps from n to 1 over δx / smoothing_offset at corona_radius (utils.jl:158-168)."""
        return [(self.corona_radius - 1.25, 0.0), (self.corona_radius, 2.5e-4), (self.corona_radius + 1.25, 0.0)]


@dataclass(frozen=True)
class NoZMetric(AbstractStaticAxisSymmetric):
    """NoZMetric(M, a, ϵ) -- src/metrics/noz-metric.jl:7-66 (a Kerr deformation without reflection symmetry
    about the equatorial plane).  Circular orbits leave the plane θ = π/2 there, so its `isco` and circular
    velocities (noz-metric.jl:68-120) are not provided; tracing and every geometric callback work."""

    M: float = 1.0
    a: float = 0.0
    ϵ: float = 0.0
    metric_id = GR_METRIC_NOZ

    def abi_params(self):
        return [self.M, self.a, self.ϵ]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, y):
        M, a = self.M, self.a
        a2 = a * a
        s2, y2 = s * s, y * y
        eps = self.ϵ * M * a * y
        S = r * r + a2 * y2
        tMr = 2.0 * M * r
        D = S * S + (r * r - tMr + a2 * y2) * eps
        Se = S + eps
        big = r ** 4 + a2 * a2 * y2 + r * r * (a2 + a2 * y2 + eps) + a2 * eps + tMr * (a2 - a2 * y2 - eps)
        return (-1.0 + tMr * S / D, Se / (r * r - tMr + a2), (Se / (1.0 - y2)) * s2, (1.0 - y2) * Se * big / D,
                -(tMr * a * (1.0 - y2) * Se) / D)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        raise NotImplementedError("NoZMetric: circular orbits leave the equatorial plane (noz-metric.jl:68-120)")


class TabulatedMetric(AbstractStaticAxisSymmetric):
    """ANY static, axis-symmetric metric on the device: the reference's plugin contract -- a struct `<: AbstractStaticAxisSymmetric`
    and one method `metric_components(m, (r, θ))` (src/Gradus.jl:78-86, src/metrics/kerr-metric.jl:62-70; ForwardDiff supplies the
    Jacobian, src/tracing/method-implementations/auto-diff.jl:206-211) -- carried across the C ABI as samples.

    `source` is an `AbstractMetric` of this package (traced through the table instead of its own kernels) or any callable
    `f(r, θ) -> (g_tt, g_rr, g_θθ, g_ϕϕ, g_tϕ)`; it is called with numpy arrays where it accepts them, point by point otherwise.
    The library names the sample nodes, fits piecewise polynomials (GR_METRIC_TABULATED, ABI 8: total degree 5) and reports its own
    error estimates; (m_r, n_theta) grow by half until the estimates are below `tol` (value) and `dtol` (derivatives).  The defaults
    -- a (24, 96) grid, estimates of 6e-11 / 3e-8 / 8e-8 for Kerr (measured: 6e-12 / 3e-9 / 2e-9), 43 MB -- leave an image 1e-11
    (median) / 5e-10 (99.9 %) from the fused kernel's.  A coarser table is faster to fit and 5 % faster to trace ((16, 64): 19 MB,
    image 1e-10 / 3e-9) but its jumps across patch edges (1e-7 in the derivatives) are what a tolerance of 1e-11, or a difference
    quotient of two traces, then resolves instead of the metric.

    `breaks`: radii where the metric's functions change form -- `(radius, scale)` pairs or bare radii (scale 0: a kink or a jump
    AT the radius; scale > 0: a smooth feature of that width centred there, resolved by patches that shrink geometrically towards it
    from both sides).  No patch straddles a break.  A source with a `break_radii()` method (KerrDarkMatter, KerrRefractive) names
    its own.  Radii may be negative (`r_min < 0`: a chart through a wormhole's throat, with a break at 0 of the throat's scale).

    `inner_radius` (the event horizon: the chart's inner radius is 1.01 of it, charts.jl:9-23; the plunging region is traced down
    to 1.000001 of it, orbit-solving.jl:162, which is where the table starts) and `isco` default to the source's; a bare callable
    must bring `inner_radius`, and its ISCO is found on the table's own derivatives.  The table covers [r_min, r_max]; a chart,
    observer or source beyond `r_max` makes it grow (`cover`), the library refuses what lies outside.
    """

    metric_id = GR_METRIC_TABULATED

    def __init__(self, source, *, inner_radius=None, isco=None, r_min=None, r_max=12000.0, r0=None, m_r=24, n_theta=96,
                 tol=1e-10, dtol=1e-7, max_refinements=3, closest_approach=1.01, pole_factor=True, strict=True, breaks=None):
        import warnings

        self.source = source
        self._f = source.metric_components if isinstance(source, AbstractMetric) else source
        if inner_radius is None:
            if not isinstance(source, AbstractMetric):
                raise ValueError("a callable metric needs `inner_radius` (the event horizon radius)")
            inner_radius = source.inner_radius()
        if breaks is None and hasattr(source, "break_radii"):
            breaks = source.break_radii()
        self.breaks = [(float(b[0]), float(b[1])) if np.ndim(b) else (float(b), 0.0) for b in (breaks or [])]
        # The table represents the metric by polynomials: a pole of g_rr INSIDE its radial range -- an `inner_radius` that lies
        # within the outermost horizon, as the reference's formulas for the dilaton-axion and Kerr-dark-matter metrics do --
        # cannot be fitted (estimates of order 1 .. 1000: rays near it would stall or scatter).  Direct evaluation carries such
        # rays across the pole in one form or another; here the range starts outside the OUTERMOST sign change of g_rr instead.
        outer = self._outermost_horizon(float(inner_radius), float(r_max)) if (r_min is None or r_min >= inner_radius) else None
        if outer is not None and r_min is None and r0 is None:
            warnings.warn(f"TabulatedMetric: g_rr changes sign at r = {outer:.6g}, outside inner_radius = {float(inner_radius):.6g}; the "
                          "table (and the chart's inner boundary) start there", stacklevel=2)
            inner_radius = outer
        self._inner_radius = float(inner_radius)
        self._isco = isco
        # The table starts a hair inside the chart's inner radius (closest_approach = 1.01 of the horizon, charts.jl:55), and its
        # radial octaves count from a point just inside the horizon -- the pole of g_rr, towards which the patches shrink.
        # (The reference traces the plunging region down to 1.000001 of the horizon, orbit-solving.jl:162.  No image ray gets below
        # the chart's 1.01, and within 1e-6 of the horizon double precision leaves g_rr itself with 1e-9 of rounding noise -- Δ is a
        # difference of O(1) terms there: the plunge of a tabulated metric is traced as far as its table reaches,
        # special_radii.interpolate_plunging_velocities; `closest_approach` or `cover` take the table lower if a chart needs it.)
        rh = self._inner_radius
        self.r_min = float(r_min) if r_min is not None else (rh * (1.0 + 0.9 * (closest_approach - 1.0)) if rh > 0 else rh)
        self._r0_given = r0 is not None
        self.r0 = float(r0) if r0 is not None else self._default_r0(self.r_min)
        self.r_max = float(r_max)
        self._fit_args = dict(m_r=int(m_r), n_theta=int(n_theta), tol=float(tol), dtol=float(dtol), max_refinements=int(max_refinements),
                              pole_factor=pole_factor, strict=bool(strict))
        self._build()

    def _default_r0(self, r_min):
        rh = self._inner_radius
        if rh > 0 and r_min > rh:
            return rh - 0.1 * (r_min - rh)               # the pole of g_rr lies a tenth of the first patch's distance behind r0
        return r_min - max(1.0, abs(r_min))              # (no horizon above r_min -- a wormhole's throat, flat space: nothing to shrink towards)

    def cover(self, r_inner, r_outer):
        """Make the table contain [r_inner, r_outer] (a chart, an observer far out): refit on a larger range when it does not --
        polynomials do not extrapolate, and the library refuses a chart that leaves the table."""
        lo, hi = self.r_min, self.r_max
        if r_outer > hi * (1.0 + 1e-9):
            hi = 2.0 * float(r_outer)
        if r_inner < lo * (1.0 - 1e-9) - 1e-12:
            rh = self._inner_radius
            if rh > 0 and not r_inner > rh:
                raise ValueError(f"TabulatedMetric: a chart that reaches r = {r_inner:.9g} crosses the horizon at {rh:.9g}")
            lo = float(r_inner) - (1e-3 * (float(r_inner) - rh) if rh > 0 else 1e-9 * max(1.0, abs(float(r_inner))))
        if (lo, hi) != (self.r_min, self.r_max):
            self.r_min, self.r_max = lo, hi
            if not self._r0_given:
                self.r0 = self._default_r0(lo)
            self._build()
        return self

    def _plan(self, m_r, n_theta):
        from . import _lib

        L = _lib.load()
        grid = _lib.gr_metric_grid()
        inside = [b for b in self.breaks if self.r_min < b[0] < self.r_max]
        if inside:
            arr = (_lib.gr_metric_break * len(inside))(*[_lib.gr_metric_break(b[0], b[1]) for b in inside])
            _lib.check(L.gr_metric_grid_plan_breaks(self.r_min, self.r_max, self.r0, int(m_r), int(n_theta), len(inside), arr, grid))
        else:
            _lib.check(L.gr_metric_grid_plan(self.r_min, self.r_max, self.r0, int(m_r), int(n_theta), grid))
        return grid

    def _build(self):
        import warnings

        from . import _lib

        a = self._fit_args
        m_r, n_theta, tol, dtol, strict = a["m_r"], a["n_theta"], a["tol"], a["dtol"], a["strict"]
        L = _lib.load()
        miss = lambda e_: max(e_[0] / tol, e_[1] / dtol, e_[2] / dtol)
        # the form g_ϕϕ and g_tϕ are stored in (gr_metric_grid.pole_factor): 1 = divided by sin²θ; tried in this order when that fails
        forms = [1, 2, 0] if a["pole_factor"] is True else [int(a["pole_factor"])]
        form = forms[0]
        self.errors = None
        for attempt in range(a["max_refinements"] + 1):
            grid = self._plan(m_r, n_theta)
            rn, tn = np.empty(grid.n_r_nodes), np.empty(grid.n_theta_nodes)
            _lib.check(L.gr_metric_grid_nodes(grid, rn.ctypes.data, tn.ctypes.data))
            samples = self._sample(rn, tn)

            def fit(form_):
                g = _lib.gr_metric_grid.from_buffer_copy(grid)
                g.pole_factor = form_
                t, e = np.empty(g.table_doubles), (ctypes.c_double * 3)()
                _lib.check(L.gr_metric_table_fit(g, samples.ctypes.data, t.ctypes.data, e))
                return g, t, tuple(e)

            g_, table, err = fit(form)
            if attempt == 0 and len(forms) > 1 and miss(err) > 1e4:
                # g_ϕϕ / sin²θ is not smooth on the axis of every metric.  With an axion charge (the dilaton-axion metric, β != 0)
                # g_ϕϕ and g_tϕ do not vanish there: form 2 takes their limits on the two poles out first, K_m(r) + K_d(r) cos θ, and
                # divides the rest.  The reference's Morris-Thorne g_ϕϕ ∝ sin θ is smooth as it is (form 0).  Same samples.
                for alt in forms[1:]:
                    g2, t2, e2 = fit(alt)
                    if miss(e2) < 1e-2 * miss(err):
                        g_, table, err, form = g2, t2, e2, alt
                        break
            previous = self.errors
            self.grid, self.table, self.errors = g_, table, err
            if err[0] <= tol and err[1] <= dtol and err[2] <= dtol:
                break
            # a fit of degree p gains 1.5^(p+1) = 11 per refinement of a smooth function's patches; one that gains less than 3 is looking
            # at a kink or a pole, and the refinements left (2.25x the samples each) would not close a gap of 1000
            if previous is not None and miss(self.errors) > miss(previous) / 3.0 and miss(self.errors) > 1e3:
                break
            # refine the direction(s) whose derivative estimate is worse: half as many patches again (a degree-5 fit gains 11x from that)
            if err[1] > dtol or err[0] > tol:
                m_r = (3 * m_r + 1) // 2
            if err[2] > dtol or err[0] > tol:
                n_theta = (3 * n_theta + 1) // 2
        self.m_r, self.n_theta = int(self.grid.m_r), int(self.grid.n_theta)
        self._segs = [self.grid.seg[k] for k in range(self.grid.n_seg)]
        e = self.errors
        if e[0] > 100.0 * tol or e[1] > 100.0 * dtol or e[2] > 100.0 * dtol:
            msg = (f"TabulatedMetric: the fit's error estimates (value {e[0]:.2g}, ∂r {e[1]:.2g}, ∂θ {e[2]:.2g}; asked {tol:.2g}, {dtol:.2g}) did not "
                   f"come down on the grid ({self.m_r}, {self.n_theta}): the metric is not smooth on r in [{self.r_min:.6g}, {self.r_max:.6g}] -- a "
                   "horizon inside the range (raise `inner_radius` / `r_min`), a kink or a jump in one of its functions (name the radius "
                   "in `breaks`), or NaNs")
            if strict and not (e[0] <= 1e-7 and e[1] <= 1e-4 and e[2] <= 1e-4):
                raise ValueError(msg + "; strict=False traces through the table as it is")
            warnings.warn(msg, stacklevel=3)

    def _outermost_horizon(self, r_in, r_max):
        """The largest r in (r_in, r_max) where g_rr changes sign (or stops being finite) on the equator or near the axis, refined
        by bisection; None if g_rr > 0 all the way."""
        found = None
        rs = r_in + np.geomspace(1e-6 * max(r_in, 1.0), max(r_max - r_in, 1.0), 400)
        for th in (0.5 * math.pi, 0.3):
            try:
                grr = np.array([float(np.asarray(self._f(float(r), th)[1])) for r in rs])
            except Exception:      # a callable that cannot be evaluated below its own horizon: nothing to look for
                return None
            bad = ~(np.isfinite(grr) & (grr > 0.0))
            if not bad.any():
                continue
            k = int(np.nonzero(bad)[0][-1])
            if k + 1 >= rs.size:
                continue
            lo, hi = rs[k], rs[k + 1]
            for _ in range(80):
                mid = 0.5 * (lo + hi)
                g = float(np.asarray(self._f(mid, th)[1]))
                if np.isfinite(g) and g > 0.0:
                    hi = mid
                else:
                    lo = mid
            found = hi if found is None else max(found, hi)
        return found

    def _sample(self, rn, tn):
        """metric_components on the tensor grid of nodes -> array [n_r, n_θ, 5]."""
        R, T = np.meshgrid(rn, tn, indexing="ij")
        try:
            if isinstance(self.source, AbstractMetric) and hasattr(self.source, "_components"):
                # (a catalogue type's metric_components takes one point -- math.sin; its formula takes arrays)
                g = self.source._components(R, np.sin(T), np.cos(T))
            else:
                g = self._f(R, T)
            out = np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), R.shape) for c in g], axis=-1)
        except (TypeError, ValueError):
            out = np.empty(R.shape + (5,))
            for a in range(R.shape[0]):
                for b in range(R.shape[1]):
                    out[a, b] = self._f(float(R[a, b]), float(T[a, b]))
        return np.ascontiguousarray(out, dtype=np.float64)

    def abi_params(self):
        return []

    def metric_components(self, r, theta):
        return tuple(self._f(r, theta))

    def _components(self, r, s, c):
        """(r may be a special_radii.Jet: the generic ISCO / plunging set-up differentiates through here)"""
        if isinstance(self.source, AbstractMetric) and hasattr(self.source, "_components"):
            return self.source._components(r, s, c)
        from .special_radii import Jet

        if not isinstance(r, Jet):
            # values only (disc kinematics of the host reductions, on arrays of points): the callable itself, for any shapes
            th = np.arctan2(s, c)
            try:
                return tuple(np.asarray(x, dtype=np.float64) if np.ndim(x) else float(x) for x in self._f(r, th))
            except (TypeError, ValueError):
                rr, tt = np.broadcast_arrays(np.asarray(r, dtype=np.float64), np.asarray(th, dtype=np.float64))
                out = np.array([self._f(float(a), float(b)) for a, b in zip(rr.ravel(), tt.ravel())]).reshape(rr.shape + (5,))
                return tuple(out[..., k] for k in range(5))
        return self._table_components(r, math.atan2(s, c))

    def table_jacobian(self, r, theta):
        """(g, ∂r g, ∂θ g) from the table, by the arithmetic of the kernels (gr_metric_table_eval)."""
        import ctypes as C

        from . import _lib

        g, dr, dth = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
        _lib.check(_lib.load().gr_metric_table_eval(self.table.ctypes.data, self.table.size, float(r), float(theta), g, dr, dth))
        return np.array(g), np.array(dr), np.array(dth)

    def _rows_of(self, rv):
        """Radial row of every radius of an array, and the affine map u = A r + B of that row (gr_tab::locate_row, vectorised)."""
        rv = np.asarray(rv, dtype=np.float64)
        m_r = self.m_r
        sidx = np.zeros(rv.shape, dtype=np.int64)
        for q in range(1, len(self._segs)):
            sidx += rv >= self._segs[q].r_lo
        row, A, B = np.empty(rv.shape, dtype=np.int64), np.empty(rv.shape), np.empty(rv.shape)
        for q, sg in enumerate(self._segs):
            sel = sidx == q
            if not sel.any():
                continue
            d = float(sg.dir)
            x = d * (rv[sel] - sg.anchor)
            x = np.clip(x, 0.0, 2.0 ** (sg.e_hi + 2))
            lin = (x < sg.xmin) if sg.core else np.zeros(x.shape, dtype=bool)
            xx = np.where(lin, sg.xmin, np.maximum(x, sg.xmin))
            e = np.clip(np.floor(np.log2(xx)).astype(np.int64), sg.e_lo, sg.e_hi)
            e = np.where(lin, sg.e_lo, e)
            sc = np.exp2(-e.astype(np.float64))
            f = np.where(lin, x * sc + 1.0, xx * sc)
            j = np.clip(((f - 1.0) * m_r).astype(np.int64), 0, m_r - 1)
            octv = np.where(lin, 0, e - sg.e_lo + sg.core)
            row[sel] = sg.first_row + octv * m_r + j
            # u = 2 ((f - 1) m_r - j) - 1 with f = x 2^-e (+ 1 in the core), x = d (r - anchor)
            a_ = 2.0 * m_r * sc * d
            A[sel] = a_
            B[sel] = -a_ * sg.anchor + np.where(lin, 0.0, -2.0 * m_r) - 2.0 * j - 1.0
        return row, A, B

    def _table_components(self, r, theta):
        """The five components from the table with `r` a float or a Jet (value, d/dr, d²/dr²): the patch polynomial is
        evaluated in the number type of r, so its exact derivatives come along."""
        from .special_radii import Jet

        rv = r.v if isinstance(r, Jet) else r
        if np.ndim(rv):
            # an array of radii (the generic ISCO's downward scan): every radius in its own patch -- grouped, one evaluation per
            # group on that group's slice of r
            rv = np.asarray(rv, dtype=np.float64)
            key, _, _ = self._rows_of(rv)
            outs = [np.empty(rv.shape) for _ in range(5)]
            jets = isinstance(r, Jet)
            if jets:
                o_d, o_dd = [np.empty(rv.shape) for _ in range(5)], [np.empty(rv.shape) for _ in range(5)]
            pick = lambda z, sel: z[sel] if isinstance(z, np.ndarray) else z
            for kk in np.unique(key):
                sel = key == kk
                sub = Jet(rv[sel], pick(r.d, sel), pick(r.dd, sel)) if jets else rv[sel]
                res = self._table_components_one_patch(sub, theta)
                for c in range(5):
                    if jets:
                        z = Jet.lift(res[c])
                        outs[c][sel], o_d[c][sel], o_dd[c][sel] = z.v, z.d, z.dd
                    else:
                        outs[c][sel] = res[c]
            return tuple(Jet(outs[c], o_d[c], o_dd[c]) for c in range(5)) if jets else tuple(outs)
        return self._table_components_one_patch(r, theta)

    def _table_components_one_patch(self, r, theta):
        """(r: a float, a Jet, or an array / Jet of arrays that lies within ONE radial row)"""
        from .special_radii import Jet

        rv = r.v if isinstance(r, Jet) else r
        t = self.table
        n_th = int(t[6])
        w = abs(((theta + math.pi) % (2.0 * math.pi)) - math.pi)
        y = w * n_th / math.pi
        it = min(int(y), n_th - 1)
        v = 2.0 * (y - it) - 1.0
        row, A, B = self._rows_of(np.atleast_1d(np.asarray(rv, dtype=np.float64))[:1])
        row, A, B = int(row[0]), float(A[0]), float(B[0])
        u = r * A + B                                    # (the Jet carries du/dr)
        p, stride, form = int(t[1]), int(t[7]), int(t[14])          # H_DEGREE, H_STRIDE, H_POLE_FACTOR
        nc = (p + 1) * (p + 2) // 2
        base = int(t[18]) + (row * n_th + it) * stride   # H_PATCH_OFF
        out = []
        for k in range(5):
            cb = base + k * nc
            acc = None
            for i in range(p, -1, -1):                   # rows i = p .. 0 are stored in this order
                n = p - i
                off = cb + n * (n + 1) // 2
                q = t[off]
                for tt in range(1, n + 1):
                    q = q * v + t[off + tt]
                acc = q if acc is None else acc * u + q
            if k >= 3 and form != 0:
                acc = acc * (math.sin(theta) ** 2)
                if form == 2:
                    ab = int(t[17]) + row * 4 * (p + 1) + 2 * (k - 3) * (p + 1)      # H_AXIS_OFF: K_m then K_d of this component
                    km, kd = t[ab], t[ab + p + 1]
                    for tt in range(1, p + 1):
                        km = km * u + t[ab + tt]
                        kd = kd * u + t[ab + p + 1 + tt]
                    acc = acc + km + kd * math.cos(theta)
            out.append(acc)
        return tuple(out)

    def inner_radius(self):
        return self._inner_radius

    def isco(self):
        if self._isco is not None:
            return float(self._isco)
        if isinstance(self.source, AbstractMetric):
            return self.source.isco()
        from .special_radii import generic_isco

        return generic_isco(self)


def kerr_isco(M, a):
    """__BoyerLindquistFO.isco -- Bardeen et al. (1972) eq. 2.21; kerr-metric-first-order.jl:297-337."""
    x = a / M
    Z1 = 1.0 + np.cbrt(1.0 - x * x) * (np.cbrt(1.0 + x) + np.cbrt(1.0 - x))
    Z2 = math.sqrt(3.0 * x * x + Z1 * Z1)
    s = math.sqrt((3.0 - Z1) * (3.0 + Z1 + 2.0 * Z2))
    return float(M * (3.0 + Z2 - s) if a > 0.0 else M * (3.0 + Z2 + s))


def inner_radius(m):
    return m.inner_radius()


def isco(m):
    return m.isco()
