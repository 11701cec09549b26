"""Circular orbits found by tracing -- src/orbits/orbit-solving.jl:1-97.

    trace_single_orbit(m, r, vϕ)                 :1-6    a μ = 1 geodesic launched tangentially in the equatorial plane
    measure_stability(m, r, vϕ)                  :8-13   mean squared relative radial excursion over its saved steps
    solve_equatorial_circular_orbit(m, r | rs)   :15-88  the vϕ that minimises it (Optim.GoldenSection)
    trace_equatorial_circular_orbit(m, r | rs)   :90-97

An independent check of CircularOrbits.vϕ (the reference's test/smoke-tests/circular-orbits.jl sums the
optimiser's result over r = 6:0.5:10 and compares it with recorded values at atol 1e-6).

MI355X-first: the reference solves the radii one after another (each golden-section search starts from a window
around the previous radius' answer) and traces one orbit per objective evaluation.  Here every radius advances
in lock-step on the full bracket: one golden-section iteration of ALL radii is one `gr_trace_paths` launch.
`Optim.GoldenSection` (third party) is restated from its published algorithm, with its default tolerances.
"""
from __future__ import annotations

import math

import numpy as np

from .tracing import tracegeodesic_path, tracegeodesic_paths

_GOLDEN = 0.5 * (3.0 - math.sqrt(5.0))


def _orbit_inputs(r, vϕ, θ0):
    r, vϕ = np.broadcast_arrays(np.asarray(r, dtype=np.float64), np.asarray(vϕ, dtype=np.float64))
    n = r.size
    x = np.zeros((n, 4))
    x[:, 1], x[:, 2] = r.ravel(), θ0
    v = np.zeros((n, 4))
    v[:, 3] = vϕ.ravel()
    return x, v


def trace_single_orbit(m, r, vϕ, *, max_time=300.0, μ=1.0, θ0=math.pi / 2, **tracer_args):
    """trace_single_orbit (:1-6): the saved path of the geodesic from (0, r, θ₀, 0) with velocity (·, 0, 0, vϕ)."""
    x, v = _orbit_inputs(r, vϕ, θ0)
    return tracegeodesic_path(m, x[0], v[0], (0.0, float(max_time)), μ=μ, **tracer_args)


def measure_stability(m, r, vϕ, *, max_time=300.0, μ=1.0, θ0=math.pi / 2, **tracer_args):
    """measure_stability (:8-13) for arrays of (r, vϕ): every orbit in one launch."""
    x, v = _orbit_inputs(r, vϕ, θ0)
    paths = tracegeodesic_paths(m, x, v, (0.0, float(max_time)), μ=μ, **tracer_args)
    return np.array([np.sum(((p.x[:, 1] - x[i, 1]) / x[i, 1]) ** 2) / p.x.shape[0] for i, p in enumerate(paths)])


def _golden_section_minimizers(f, lower, upper, rel_tol=math.sqrt(np.finfo(np.float64).eps), abs_tol=np.finfo(np.float64).eps,
                               iterations=1000):
    """Optim.optimize(f, lower, upper, GoldenSection()) for independent problems in lock-step (converged ones keep
    their point and are evaluated along for the ride).  Returns Optim.minimizer of each."""
    lower, upper = np.array(lower, dtype=np.float64), np.array(upper, dtype=np.float64)
    x = lower + _GOLDEN * (upper - lower)
    fx = f(x)
    done = np.zeros(x.size, dtype=bool)
    for _ in range(iterations):
        mid = 0.5 * (upper + lower)
        tol = rel_tol * np.abs(x) + abs_tol
        done |= np.abs(x - mid) <= 2.0 * tol - 0.5 * (upper - lower)
        if done.all():
            break
        right = (upper - x) > (x - lower)
        nx = np.where(right, x + _GOLDEN * (upper - x), x - _GOLDEN * (x - lower))
        nx = np.where(done, x, nx)
        nf = f(nx)
        better = (nf < fx) & ~done
        worse = ~better & ~done
        lower = np.where(right & better, x, np.where(~right & worse, nx, lower))
        upper = np.where(right & worse, nx, np.where(~right & better, x, upper))
        x = np.where(better, nx, x)
        fx = np.where(better, nf, fx)
    return x


def solve_equatorial_circular_orbit(m, r, *, lower_bound=0.0, upper_bound=1.0, **tracer_args):
    """solve_equatorial_circular_orbit(m, r) / (m, r_range) (:15-88): vϕ of the circular orbit at each radius."""
    scalar = np.ndim(r) == 0
    rs = np.atleast_1d(np.asarray(r, dtype=np.float64))
    vϕ = _golden_section_minimizers(lambda v: measure_stability(m, rs, v, **tracer_args), np.full(rs.size, float(lower_bound)),
                                    np.full(rs.size, float(upper_bound)))
    return float(vϕ[0]) if scalar else vϕ


def trace_equatorial_circular_orbit(m, r, **kwargs):
    """trace_equatorial_circular_orbit (:90-97): the saved paths of the solved orbits."""
    scalar = np.ndim(r) == 0
    rs = np.atleast_1d(np.asarray(r, dtype=np.float64))
    vϕ = solve_equatorial_circular_orbit(m, rs, **kwargs)
    kw = {k: v for k, v in kwargs.items() if k not in ("lower_bound", "upper_bound")}
    max_time, μ, θ0 = kw.pop("max_time", 300.0), kw.pop("μ", 1.0), kw.pop("θ0", math.pi / 2)
    x, v = _orbit_inputs(rs, vϕ, θ0)
    paths = tracegeodesic_paths(m, x, v, (0.0, float(max_time)), μ=μ, **kw)
    return paths[0] if scalar else paths
