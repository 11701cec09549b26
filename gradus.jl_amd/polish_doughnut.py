"""PolishDoughnut(m; rₖ, n) -- src/geometry/discs/polish-doughnut.jl.

A pressure-supported torus whose surface is the isobar through the innermost radius (where
dE/dr = 0 for the power-law rotation Ω = Ω_K(ρ) (rₖ/ρ)^n).  The reference integrates the isobar
with OrdinaryDiffEq's Tsit5 (default tolerances abstol 1e-6 / reltol 1e-3, dtmax = 5e-2) and
wraps the saved steps in a linear interpolation; both are restated here (SURVEY App. A for the
stepper: same tableau, error norm, PI controller and initial step as the geodesic integrator).
The resulting cross-section is then sampled onto the device like any `ThickDisc(f)`.
One-off host set-up; nothing here is on the per-ray path."""
from __future__ import annotations

import math

import numpy as np

from .geometry import ThickDisc
from .special_radii import Jet

# Tsit5 (Tsitouras 2011); same coefficients as gr_device.hpp `Ts`
_C = [0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0]
_A = [
    [],
    [0.161],
    [-0.008480655492356989, 0.335480655492357],
    [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
    [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
    [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
    [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774],
]
_BT = [-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
       0.5823571654525552, -0.45808210592918697, 0.015151515151515152]


def tsit5_solve(f, u0, t0, t1, *, abstol=1e-6, reltol=1e-3, dtmax=None, terminate=None, maxiters=1_000_000):
    """solve(ODEProblem(f, u0, (t0, t1)), Tsit5(); dtmax, callback = DiscreteCallback(terminate, terminate!))
    with OrdinaryDiffEq's defaults, saving every accepted step (and the duplicate a terminating
    DiscreteCallback appends, save_positions = (true, true)).  Returns the list of saved states."""
    u = np.asarray(u0, dtype=np.float64)
    n = u.size
    dtmax = abs(t1 - t0) if dtmax is None else dtmax
    rms = lambda v: math.sqrt(float(np.sum(v * v)) / n)
    # initial step (App. A.4)
    sk = abstol + np.abs(u) * reltol
    f0 = np.asarray(f(u), dtype=np.float64)
    d0, d1 = rms(u / sk), rms(f0 / sk)
    dt0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    dt0 = min(dt0, dtmax)
    f1 = np.asarray(f(u + dt0 * f0), dtype=np.float64)
    d2 = rms((f1 - f0) / sk) / dt0
    dm = max(d1, d2)
    dt1 = max(1e-6, 1e-3 * dt0) if dm <= 1e-15 else 10.0 ** (-(2.0 + math.log10(dm)) / 5.0)
    dt = min(100.0 * dt0, dt1, dtmax)
    t, qold = t0, 1e-4
    k1 = f0
    saved = [u.copy()]
    for _ in range(maxiters):
        h = min(dt, t1 - t)
        ks = [k1]
        for s in range(1, 7):
            us = u + h * sum(_A[s][j] * ks[j] for j in range(s))
            ks.append(np.asarray(f(us), dtype=np.float64))
        unew = us                                       # stage 7 argument is the new state (FSAL)
        err = h * sum(_BT[j] * ks[j] for j in range(7))
        EEst = rms(err / (abstol + np.maximum(np.abs(u), np.abs(unew)) * reltol))
        if EEst == 0.0:
            q11, q = 0.0, 0.1
        else:
            q11 = EEst ** 0.14
            q = min(5.0, max(0.1, q11 / qold ** 0.08 / 0.9))
        if EEst <= 1.0:
            t = t + h
            if abs(t - t1) < 100 * np.finfo(float).eps * max(abs(t), abs(t1)):
                t = t1
            u, k1 = unew, ks[6]
            qold = max(EEst, 1e-4)
            dt = min(dtmax, h / q)
            saved.append(u.copy())
            if terminate is not None and terminate(u):
                saved.append(u.copy())
                break
            if not (t < t1):
                break
        else:
            dt = h / min(5.0, q11 / 0.9)
    return saved


class PolishDoughnut:
    """PolishDoughnut(m; rₖ = 12.0, n = 0.21, init_r = 5.0): inner_radius, outer_radius and the
    cross-section z(r) of the torus (polish-doughnut.jl:100-123)."""

    def __init__(self, m, rₖ=12.0, n=0.21, init_r=5.0, λ_max=40.0, dtmax=5e-2):
        self.metric, self.rₖ, self.n = m, float(rₖ), float(n)
        self.inner_radius = innermost_radius(m, self.rₖ, self.n, init_r=init_r)
        self.r, self.z = isobar(m, self.inner_radius, self.rₖ, self.n, λ_max=λ_max, dtmax=dtmax)
        self.outer_radius = float(np.max(self.r))

    def cross_section(self, ρ):
        ρ = np.asarray(ρ, dtype=np.float64)
        inside = (self.inner_radius <= ρ) & (ρ <= self.outer_radius)
        # DataInterpolations.LinearInterpolation(z, r): knots are the saved steps
        idx = np.clip(np.searchsorted(self.r, ρ, side="right") - 1, 0, self.r.size - 2)
        w = (ρ - self.r[idx]) / (self.r[idx + 1] - self.r[idx])
        return np.where(inside, self.z[idx] + w * (self.z[idx + 1] - self.z[idx]), 0.0)

    def thick_disc(self, samples=16384):
        """The same surface as a device geometry: `ThickDisc(f)` sampled on a uniform ρ grid."""
        return ThickDisc(lambda ρ: float(self.cross_section(ρ)), ρ_range=(self.inner_radius, self.outer_radius),
                         samples=samples)


def _omega_K(m, ρ):
    """CircularOrbits.Ω(m, (ρ, π/2)) = _Ω_analytic(∂_r g) (circular-orbits.jl:11-24)"""
    g = m._components(Jet(ρ, 1.0, 0.0), 1.0, 0.0)
    dg = [Jet.lift(c).d for c in g]
    return -(dg[4] - math.sqrt(dg[4] * dg[4] - dg[0] * dg[3])) / dg[3]


def _orbital_energy_jet(m, r, rₖ, n):
    """orbital_energy at θ = π/2 with its r-derivative (polish-doughnut.jl:21-29)"""
    g = [Jet.lift(c) for c in m._components(Jet(r, 1.0, 0.0), 1.0, 0.0)]
    dg = [Jet(c.d, c.dd) for c in g]
    g0 = [Jet(c.v, c.d) for c in g]
    ΩK = -(dg[4] - (dg[4] * dg[4] - dg[0] * dg[3]).sqrt()) / dg[3]
    rj = Jet(r, 1.0)
    x = Jet(rₖ) / rj
    pw = Jet(x.v ** n, n * x.v ** (n - 1.0) * x.d)          # (rₖ/r)^n to first order
    Ω = ΩK * pw
    E = -(g0[0] + g0[4] * Ω) / (-(g0[0]) - 2.0 * (g0[4] * Ω) - g0[3] * (Ω * Ω)).sqrt()
    return E.v, E.d


def innermost_radius(m, rₖ, n, init_r=5.0):
    """Root of dE/dr nearest to `init_r` (the reference runs Newton from there with ForwardDiff's
    second derivative; a bracketing search finds the same root)."""
    def dE(r):
        with np.errstate(all="ignore"):
            return _orbital_energy_jet(m, r, rₖ, n)[1]

    f0 = dE(init_r)
    step = 0.05
    lo = hi = init_r
    flo = fhi = f0
    for _ in range(2000):
        lo -= step
        if lo > m.inner_radius():
            flo = dE(lo)
            if flo == flo and (flo > 0) != (f0 > 0):
                a, b, fa = lo, lo + step, flo
                break
        hi += step
        fhi = dE(hi)
        if fhi == fhi and (fhi > 0) != (f0 > 0):
            a, b, fa = hi - step, hi, dE(hi - step)
            break
    else:
        raise RuntimeError("no innermost radius found")
    for _ in range(200):
        mid = 0.5 * (a + b)
        if not (a < mid < b):
            break
        fm = dE(mid)
        if (fm > 0) == (fa > 0):
            a, fa = mid, fm
        else:
            b = mid
    return 0.5 * (a + b)


def isobar(m, inner_radius, rₖ, n, λ_max=40.0, dtmax=5e-2):
    """polish-doughnut.jl:61-97 (Kerr only, like the reference's isobar_differential)"""
    M, a = m.M, m.a

    def rhs(u):
        r, θ = u
        s, c = math.sin(θ), math.cos(θ)
        Ω = _omega_K(m, r * s) * (rₖ / (r * s)) ** n
        iΩ = 1.0 / Ω
        Σ = r * r + a * a * c * c
        Δ = r * r + a * a - 2.0 * M * r
        ψ1 = M * ((Σ - 2.0 * r * r) / (Σ * Σ)) * (iΩ - a * s) ** 2 + r * s * s
        ψ2 = math.sin(2.0 * θ) * ((M * r / (Σ * Σ)) * (a * iΩ - (r * r + a * a)) ** 2 + Δ / 2.0)
        d = 1.0 / (math.sqrt(Δ * ψ1 * ψ1 + ψ2 * ψ2) * math.sqrt(1.0 / (Δ / Σ)))
        return np.array([ψ2 * d, -ψ1 * d])

    sol = tsit5_solve(rhs, [inner_radius, math.pi / 2], 0.0, λ_max, dtmax=dtmax,
                      terminate=lambda u: u[0] * math.cos(u[1]) < 0)
    r = np.array([u[0] for u in sol])
    z = np.array([u[0] * math.cos(u[1]) for u in sol])
    keep = z > 0
    return r[keep], z[keep]
