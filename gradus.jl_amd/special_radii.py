"""Host-side, once-per-render set-up for non-Kerr redshift: generic ISCO and the plunging table.

Reference: src/special-radii.jl:14-60 (isco root find), src/orbits/circular-orbits.jl:11-48,128-146
(Ω, u_t, u_ϕ, energy, plunging_fourvelocity), src/orbits/orbit-solving.jl:99-167
(PlungingInterpolation).  The reference differentiates the metric with ForwardDiff; here a small
jet type (value, d/dr, d²/dr²) plays that role.  The one plunging geodesic is traced by the device
integrator (gr_trace_path), not on the host.
"""
from __future__ import annotations

import math

import numpy as np


class Jet:
    """Truncated Taylor series in one variable: (v, d, dd)."""

    __slots__ = ("v", "d", "dd")

    def __init__(self, v, d=0.0, dd=0.0):
        # floats, or numpy arrays of one shape (vectorised evaluation over many radii)
        f = lambda z: z if isinstance(z, np.ndarray) else float(z)
        self.v, self.d, self.dd = f(v), f(d), f(dd)

    @staticmethod
    def lift(x):
        return x if isinstance(x, Jet) else Jet(x)

    def __add__(self, o):
        o = Jet.lift(o)
        return Jet(self.v + o.v, self.d + o.d, self.dd + o.dd)

    __radd__ = __add__

    def __neg__(self):
        return Jet(-self.v, -self.d, -self.dd)

    def __sub__(self, o):
        return self + (-Jet.lift(o))

    def __rsub__(self, o):
        return Jet.lift(o) - self

    def __mul__(self, o):
        o = Jet.lift(o)
        return Jet(self.v * o.v, self.d * o.v + self.v * o.d, self.dd * o.v + 2.0 * self.d * o.d + self.v * o.dd)

    __rmul__ = __mul__

    def inv(self):
        i = 1.0 / self.v
        return Jet(i, -self.d * i * i, (2.0 * self.d * self.d * i - self.dd) * i * i)

    def __truediv__(self, o):
        return self * Jet.lift(o).inv()

    def __rtruediv__(self, o):
        return Jet.lift(o) * self.inv()

    def __pow__(self, n):
        assert isinstance(n, int) and n >= 0
        out = Jet(1.0)
        for _ in range(n):
            out = out * self
        return out

    def sqrt(self):
        s = np.sqrt(self.v)
        d = 0.5 * self.d / s
        return Jet(s, d, (0.5 * self.dd - d * d) / s)


def _energy_jet(m, r):
    """CircularOrbits.energy(m, r) = -u_t with its r-derivative (circular-orbits.jl:11-48)."""
    g = m._components(Jet(r, 1.0, 0.0), 1.0, 0.0)          # θ = π/2
    # ∂_r g as first-order series (value = g', derivative = g'')
    dg = [Jet(c.d, c.dd) for c in g]
    g0 = [Jet(c.v, c.d) for c in g]
    disc = dg[4] * dg[4] - dg[0] * dg[3]
    if disc.v < 0:
        return float("nan"), float("nan")
    Om = -(dg[4] - disc.sqrt()) / dg[3]                    # _Ω_analytic, prograde
    D = g0[0] * g0[3] - g0[4] * g0[4]
    itt, ipp, itp = g0[3] / D, g0[0] / D, -g0[4] / D       # inverse_metric_components
    A = -(Om * itt - itp)
    B = Om * itp - ipp
    den = B * B * itt + 2.0 * (A * B * itp) + A * A * ipp
    sg = 1.0 if den.v > 0 else -1.0
    d = (den * sg).inv().sqrt() * (-sg)                    # -sign(den) * sqrt(inv(abs(den)))
    ut = B * d
    return -ut.v, -ut.d


def _energy_values(m, r):
    """CircularOrbits.energy(m, r) for an array of radii (values only; NaN where no circular orbit exists)."""
    with np.errstate(all="ignore"):
        g = [Jet.lift(c) for c in m._components(Jet(r, np.ones_like(r), np.zeros_like(r)), 1.0, 0.0)]
        gv = [c.v + 0.0 * r for c in g]
        dg = [c.d + 0.0 * r for c in g]
        Om = -(dg[4] - np.sqrt(dg[4] * dg[4] - dg[0] * dg[3])) / dg[3]
        D = gv[0] * gv[3] - gv[4] * gv[4]
        itt, ipp, itp = gv[3] / D, gv[0] / D, -gv[4] / D
        A = -(Om * itt - itp)
        B = Om * itp - ipp
        den = B * B * itt + 2.0 * A * B * itp + A * A * ipp
        return B * np.sign(den) * np.sqrt(1.0 / np.abs(den))          # -u_t = -B d,  d = -sign(den)/sqrt|den|


def generic_isco(m, max_upper_bound=100.0, step=0.005):
    """isco(m::AbstractStaticAxisSymmetric): find_isco_bounds then a bracketing root find of
    dE/dr (special-radii.jl:14-60).  The downward scan for the bound is evaluated for all radii at once."""
    lower = None
    n = int(math.floor((max_upper_bound - 1.0) / step + 1e-9))
    rs = max_upper_bound - step * np.arange(n + 1)
    E = _energy_values(m, rs)
    bad = ~(E == E) | (np.abs(E) > 1.0)
    if bad.any():
        lower = float(rs[int(np.argmax(bad))])
    if lower is None:
        raise RuntimeError("No boundaries for minimization could be determined. It is likely this configuration "
                           "does not have an ISCO solution.")
    lo, hi = lower, max_upper_bound
    _, dlo = _energy_jet(m, lo)
    if not (dlo == dlo):
        lo += step
        _, dlo = _energy_jet(m, lo)
    _, dhi = _energy_jet(m, hi)
    if (dlo > 0) == (dhi > 0):
        raise RuntimeError("dE/dr does not change sign on the ISCO bracket")
    for _ in range(200):
        mid = lo + 0.5 * (hi - lo)
        if not (lo < mid < hi):
            break
        _, dm = _energy_jet(m, mid)
        if (dm > 0) == (dlo > 0):
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def plunging_fourvelocity(m, r):
    """CircularOrbits.plunging_fourvelocity(m, r) -- only valid AT the ISCO (circular-orbits.jl:128-146)."""
    g = m._components(Jet(r, 1.0, 0.0), 1.0, 0.0)
    gv = [c.v for c in g]
    dg = [c.d for c in g]
    Om = -(dg[4] - math.sqrt(dg[4] * dg[4] - dg[0] * dg[3])) / dg[3]
    D = gv[0] * gv[3] - gv[4] * gv[4]
    itt, ipp, itp = gv[3] / D, gv[0] / D, -gv[4] / D
    A = -(Om * itt - itp)
    B = Om * itp - ipp
    den = B * B * itt + 2.0 * A * B * itp + A * A * ipp
    d = -math.copysign(1.0, den) * math.sqrt(1.0 / abs(den))
    ut, up = B * d, A * d
    E, L = -ut, up
    vt = itt * ut + itp * up
    vp = itp * ut + ipp * up
    nom = itt * E * E - 2.0 * itp * E * L + ipp * L * L + 1.0
    return np.array([vt, -math.sqrt(abs(nom / (-gv[1]))), 0.0, vp])


class PlungingInterpolation:
    """PlungingInterpolation(m, ...) (orbit-solving.jl:99-131): the tabulated plunge (r, v^t, v^r, v^ϕ) below
    the ISCO of `m`.  Unpacks like the 4-tuple of arrays it wraps; calling it interpolates linearly."""

    def __init__(self, m, r, vt, vr, vϕ):
        self.m, self.r, self.vt, self.vr, self.vϕ = m, r, vt, vr, vϕ

    def __iter__(self):
        return iter((self.r, self.vt, self.vr, self.vϕ))

    def __getitem__(self, i):
        return (self.r, self.vt, self.vr, self.vϕ)[i]

    def __len__(self):
        return 4

    def __call__(self, r):
        from .corona import _nan_linear_interp

        r = np.asarray(r, dtype=np.float64)
        return np.stack([_nan_linear_interp(self.r, self.vt, r), _nan_linear_interp(self.r, self.vr, r),
                         np.zeros_like(r), _nan_linear_interp(self.r, self.vϕ, r)], axis=-1)


class CircularOrbits:
    """CircularOrbits.{Ω, energy, angmom, fourvelocity, plunging_fourvelocity} for prograde equatorial orbits
    of any static axis-symmetric metric (src/orbits/circular-orbits.jl:11-146); radii may be arrays."""

    @staticmethod
    def _parts(m, r):
        r = np.asarray(r, dtype=np.float64)
        with np.errstate(all="ignore"):
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
        return Om, B * d, A * d, (itt, ipp, itp)

    @staticmethod
    def Ω(m, r):
        return CircularOrbits._parts(m, r)[0]

    @staticmethod
    def energy(m, r):
        return -CircularOrbits._parts(m, r)[1]

    @staticmethod
    def angmom(m, r):
        return CircularOrbits._parts(m, r)[2]

    @staticmethod
    def fourvelocity(m, r):
        _, ut, up, (itt, ipp, itp) = CircularOrbits._parts(m, r)
        r = np.asarray(r, dtype=np.float64)
        out = np.zeros(r.shape + (4,))
        out[..., 0] = itt * ut + itp * up
        out[..., 3] = itp * ut + ipp * up
        return out

    @staticmethod
    def plunging_fourvelocity(m, r):
        return plunging_fourvelocity(m, r)


def interpolate_plunging_velocities(m, ensemble=None, max_time=50_000.0, reltol=1e-9, δr=None):
    """interpolate_plunging_velocities(m) -> (r, v^t, v^r, v^ϕ) sorted by r with the innermost
    sample dropped (PlungingInterpolation, orbit-solving.jl:99-131,137-167)."""
    from .tracing import PolarChart, tracegeodesic_path

    δr = reltol * 10 if δr is None else δr
    isco = m.isco()
    u = np.array([0.0, isco - δr, math.pi / 2, 0.0])
    v = plunging_fourvelocity(m, isco)
    inner = m.inner_radius() * 1.000001                          # chart_for_metric(m; closest_approach = 1.000001)
    if hasattr(m, "table"):
        # a tabulated metric: as far as its table reaches (metrics.TabulatedMetric: no ray of an image gets below the chart's 1.01)
        inner = max(inner, m.r_min * (1.0 + 1e-9))
    chart = PolarChart(inner, 12000.0)
    path = tracegeodesic_path(m, u, v, (0.0, max_time), μ=1.0, reltol=reltol, chart=chart, ensemble=ensemble)
    r = path.x[:, 1]
    idx = np.argsort(r, kind="stable")[1:]
    return PlungingInterpolation(m, np.ascontiguousarray(r[idx]), np.ascontiguousarray(path.v[idx, 0]),
                                 np.ascontiguousarray(path.v[idx, 1]), np.ascontiguousarray(path.v[idx, 3]))
