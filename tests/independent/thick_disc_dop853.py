#!/usr/bin/env python3
"""An integrator that shares NOTHING with the oracle or the HIP kernels, for the two thick-geometry
fingerprints of the reference's smoke test that neither reproduces
(test/smoke-tests/rendergeodesics.jl:16-30,70-96):

    ShakuraSunyaev(m) 20 x 20  recorded 34455.34416982827   oracle / HIP 34188.36
    ThickDisc(_thick_disc)     recorded 16918.69258396256   oracle / HIP 16521.16

What is independent here:
  * equations of motion: Hamilton's equations in COVARIANT momenta, H = ½ g^{μν} p_μ p_ν, for Schwarzschild
    (KerrMetric() defaults to M = 1, a = 0), hand-differentiated -- not the second-order Christoffel form;
  * stepper: scipy's DOP853 (8th order Dormand-Prince), rtol = atol = 1e-12, dense output -- not Tsit5;
  * event location: the condition is scanned on the dense output every Δλ = 0.002 and the first sign change is
    closed by brentq -- the *exact* first crossing, not the 8-samples-per-step search of DiffEqBase;
  * initial conditions: the static-observer tetrad of Schwarzschild written out by hand -- not Gram-Schmidt.

Each ray is integrated ONCE; the trajectory is then judged against several definitions of the disc condition:

  current      distance_to_disc(::AbstractThickAccretionDisc) as /root/reference has it today
               (src/geometry/discs/thick-disc.jl:60-66: height = cross_section(d, ρ), ρ = r|sinθ|;
               1 if height <= 0 else r|cosθ| - height; shakura-sunyaev.jl:28-33)
  gtol         the same minus gtol·r (the thin disc's tolerance term, src/geometry/discs.jl:7, thin-disc.jl:20-26)
  spherical    cross_section evaluated at the spherical radius r = u[2] instead of ρ -- the convention the
               ThickDisc docstring still shows (thick-disc.jl:16-27: "r = u[2]")
  spherical+gtol both

so that the script can say which definition each recorded value belongs to.  Run:  python tests/independent/thick_disc_dop853.py
(writes tests/golden/thick_disc_independent.json; ~1 min on 8 cores).
"""
from __future__ import annotations

import json
import math
import os
import sys
from multiprocessing import Pool

import numpy as np
from scipy.integrate import solve_ivp
from scipy.optimize import brentq

R_OBS, TH_OBS = 100.0, math.radians(85.0)
LAMBDA_MAX = 200.0
GTOL = 1e-2
R_INNER = 1.01 * 2.0                     # chart: 1.01 r₊, r₊ = 2M for a = 0 (charts.jl:51-58)
ISCO = 6.0
INV_ETA = 1.0 / (1.0 - math.sqrt(8.0 / 9.0))     # 1 / (1 - E_isco), CircularOrbits.energy at r = 6M
MDOT = 0.3


# ---- cross sections (smoke test :8-15, shakura-sunyaev.jl:28-33) ----
def cs_torus(rho):
    x = rho - 10.0
    return np.where((rho < 9.0) | (rho > 11.0), -1.0, np.sqrt(np.maximum(1.0 - x * x, 0.0)))


def cs_ss(rho):
    return np.where(rho < ISCO, -0.0, 3.0 * INV_ETA * MDOT * (1.0 - np.sqrt(ISCO / np.maximum(rho, 1e-300))))


def condition(cs, r, th, variant):
    rho = r * np.abs(np.sin(th))
    h = cs(r if "spherical" in variant else rho)
    c = r * np.abs(np.cos(th)) - h
    if "gtol" in variant:
        c = c - GTOL * np.abs(r)
    return np.where(h <= 0.0, 1.0, c)


VARIANTS = ("current", "gtol", "spherical", "spherical+gtol")
DISCS = {"shakura_sunyaev": cs_ss, "thick_torus": cs_torus}


# ---- Hamiltonian geodesics in Schwarzschild: y = (t, r, θ, ϕ, p_r, p_θ); p_t = -E, p_ϕ = L constant ----
def rhs(_, y, E, L):
    r, th, pr, pth = y[1], y[2], y[4], y[5]
    f = 1.0 - 2.0 / r
    fp = 2.0 / (r * r)
    s, c = math.sin(th), math.cos(th)
    r2 = r * r
    return [E / f, f * pr, pth / r2, L / (r2 * s * s),
            -0.5 * (E * E * fp / (f * f) + fp * pr * pr - 2.0 * pth * pth / (r2 * r) - 2.0 * L * L / (r2 * r * s * s)),
            L * L * c / (r2 * s * s * s)]


def initial_state(alpha, beta):
    """local_momentum (tracing/utility.jl:13-20) through the static tetrad of Schwarzschild, then the null
    constraint for v^t (constraints.jl:14-15): returns y0, E, L."""
    r, th = R_OBS, TH_OBS
    f = 1.0 - 2.0 / r
    a, b = alpha / r, beta / r
    pr_loc = -1.0 / math.sqrt(1.0 + a * a + b * b)
    vr = math.sqrt(f) * pr_loc
    vth = b * pr_loc / r
    vph = a * pr_loc / (r * math.sin(th))
    vt = math.sqrt((vr * vr / f + r * r * vth * vth + (r * math.sin(th)) ** 2 * vph * vph) / f)
    E = f * vt
    L = (r * math.sin(th)) ** 2 * vph
    return [0.0, r, th, 0.0, vr / f, r * r * vth], E, L


def trace_one(args):
    alpha, beta = args
    y0, E, L = initial_state(alpha, beta)

    def horizon(_, y, *a):
        return y[1] - R_INNER

    horizon.terminal = True
    horizon.direction = -1
    sol = solve_ivp(rhs, (0.0, LAMBDA_MAX), y0, method="DOP853", rtol=1e-12, atol=1e-12, dense_output=True,
                    events=horizon, args=(E, L))
    lam_end = float(sol.t[-1])
    captured = sol.status == 1
    # the reference's chart callback is a DiscreteCallback: the ray stops at the first STEP END inside 1.01 r₊,
    # this integrator stops at the crossing itself; steps there are ~1e-3..1e-2 long, 1e-6 of the fingerprint
    lam = np.arange(0.0, lam_end, 0.002)
    if lam.size == 0 or lam[-1] < lam_end:
        lam = np.append(lam, lam_end)
    Y = sol.sol(lam)
    r, th = Y[1], Y[2]
    out = {"lambda_end": lam_end, "captured": bool(captured), "nfev": int(sol.nfev)}
    for dname, cs in DISCS.items():
        for var in VARIANTS:
            c = condition(cs, r, th, var)
            s0 = np.sign(c[0])
            flip = np.nonzero(np.sign(c) * s0 < 0)[0]
            ev = None
            if s0 != 0 and flip.size:
                k = int(flip[0])
                lo, hi = float(lam[k - 1]), float(lam[k])

                def g(x):
                    yy = sol.sol(x)
                    return float(condition(cs, np.array([yy[1]]), np.array([yy[2]]), var)[0])

                glo, ghi = g(lo), g(hi)
                if abs(glo) == 1.0 or abs(ghi) == 1.0 or glo * ghi > 0:
                    # the condition jumps (edge of the radial range): bisect on the sign alone
                    for _ in range(60):
                        mid = 0.5 * (lo + hi)
                        if np.sign(g(mid)) == s0:
                            lo = mid
                        else:
                            hi = mid
                    ev = lo
                else:
                    ev = float(brentq(g, lo, hi, xtol=1e-13, rtol=1e-15))
            out[f"{dname}/{var}"] = ev
    return out


def pixel_grid():
    a = np.linspace(-9.5, 9.5, 20) + 1e-6
    return [(float(x), float(y)) for x in a for y in a]


def fingerprints(rays):
    """Σ λ_max over rays that stopped early (default pf = affine_time ∘ filter_early_term, rendering.jl:93-95)."""
    res = {"shadow": sum(r["lambda_end"] for r in rays if r["captured"])}
    for dname in DISCS:
        for var in VARIANTS:
            tot = 0.0
            hits = 0
            for r in rays:
                ev = r[f"{dname}/{var}"]
                if ev is not None:
                    tot += ev
                    hits += 1
                elif r["captured"]:
                    tot += r["lambda_end"]
            res[f"{dname}/{var}"] = {"fingerprint": tot, "disc_hits": hits}
    return res


RECORDED = {"shadow": 9009.452876609641, "shakura_sunyaev": 34455.34416982827, "thick_torus": 16918.69258396256,
            "thin_disc": 38412.08347901267}
BUILD = {"shakura_sunyaev": 34188.36, "thick_torus": 16521.16}       # oracle == HIP


def main():
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        rays = pool.map(trace_one, pixel_grid(), chunksize=5)
    fp = fingerprints(rays)
    out = {"generator": "tests/independent/thick_disc_dop853.py", "integrator": "scipy DOP853 rtol=atol=1e-12, Hamiltonian form",
           "recorded_by_reference": RECORDED, "oracle_and_hip": BUILD, "independent": fp,
           "per_ray_current": {d: [r[f"{d}/current"] for r in rays] for d in DISCS},
           "per_ray_lambda_end": [r["lambda_end"] for r in rays], "per_ray_captured": [r["captured"] for r in rays]}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "thick_disc_independent.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"shadow: independent {fp['shadow']:.6f}  recorded {RECORDED['shadow']:.6f}  rel {fp['shadow'] / RECORDED['shadow'] - 1:+.2e}")
    for d in DISCS:
        for var in VARIANTS:
            v = fp[f"{d}/{var}"]
            print(f"{d:16s} {var:15s} {v['fingerprint']:14.6f} hits {v['disc_hits']:3d}   vs recorded {v['fingerprint'] / RECORDED[d] - 1:+.3e}"
                  f"   vs oracle/HIP {v['fingerprint'] / BUILD[d] - 1:+.3e}")


if __name__ == "__main__":
    sys.exit(main())
