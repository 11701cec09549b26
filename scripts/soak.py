#!/usr/bin/env python3
"""Soak test: N random scenes through the C ABI against the oracle (same acceptance as
tests/test_gpu_parity.py::test_randomised_scenes_on_device_vs_oracle), any seed, all eleven metric
families, thin / datum / Shakura-Sunyaev / sampled thick / elliptical / precessing / warped discs, both kernels.  Prints failing cases.

    python scripts/soak.py [n_scenes] [seed]
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradus_jl_amd as G
from oracle import oracle

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# python scripts/soak.py N SEED CASE [tol]: replay only scene CASE (optionally at another tolerance), on the device or --
# with SOAK_HOST=1 -- on the host-compiled kernel logic (tests/host_harness.cpp), and print its parameters
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
tol_override = float(sys.argv[4]) if len(sys.argv) > 4 else None
HOST = os.environ.get("SOAK_HOST") == "1"
if HOST:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as Hh
rng = np.random.default_rng(seed)
U = lambda a, b: float(rng.uniform(a, b))
ens = None if HOST else G.EnsembleMI355X(0)

fam = [
    ("kerr", lambda: (1.0, U(-0.998, 0.998)), G.KerrMetric),
    ("johannsen", lambda: (1.0, U(0, 0.9), U(-1, 2), U(-1, 1), U(-1, 1), U(-1, 2)), G.JohannsenMetric),
    ("bumblebee", lambda: (1.0, U(0, 0.29), U(-0.5, 1)), G.BumblebeeMetric),
    ("kerr-newman", lambda: (lambda a: (1.0, a, U(0, math.sqrt(1 - a * a) * 0.95)))(U(0, 0.9)), G.KerrNewmanMetric),
    ("johannsen-psaltis", lambda: (1.0, U(0, 0.8), U(-0.5, 1)), G.JohannsenPsaltisMetric),
    ("morris-thorne", lambda: (U(0.5, 3),), G.MorrisThorneWormhole),
    ("dilaton-axion", lambda: (1.0, U(0.1, 0.8), U(-0.3, 0.3), U(0.3, 1.5)), G.DilatonAxion),
    ("spherical", lambda: (), G.SphericalMetric),
    ("kerr-dark-matter", lambda: (1.0, U(0, 0.9), U(0, 3), U(5, 30), U(5, 20)), G.KerrDarkMatter),
    ("kerr-refractive", lambda: (1.0, U(0, 0.9), U(0.9, 1.3), U(10, 30)), G.KerrRefractive),
    ("noz", lambda: (1.0, U(0, 0.9), U(-0.5, 0.5)), G.NoZMetric),
]
bad = []
tot = mis = 0
worst = 0.0
unexplained_total = 0
unexplained_err_scenes = 0
for case in range(n_scenes):
    name, gen, cls = fam[int(rng.integers(0, len(fam)))]
    params = gen()
    r_obs = float(10 ** U(1.3, 3.2))
    th = float(np.radians(U(5, 175)))
    kind = ["thin", "thin", "datum", "ss", "table", "ellipse", "precess", "warped"][int(rng.integers(0, 8))]
    if kind == "datum" and th > math.pi / 2 - 0.1:
        kind = "thin"
    rin = U(0, 8)
    rout = rin + 10 ** U(0, 2.3)
    gtol = float(10 ** U(-2.3, -1))
    tol = float(rng.choice([1e-9, 1e-7, 1e-5]))
    hemi = bool(rng.integers(0, 2))
    q = U(-1, 1) if (name == "kerr-newman" and rng.integers(0, 2)) else 0.0
    lam = U(1.2, 3) * r_obs
    lim = U(5, 60)
    W = H = 16
    m = cls(*params)
    x = np.array([0.0, r_obs, th, 0.0])
    kern = int(rng.integers(0, 2))
    if not HOST:
        ens.set("kernel", kern)
    if kind == "thin":
        d, od = G.ThinDisc(rin, rout), (rin, rout)
    elif kind == "datum":
        h = U(0, 2)
        d, od = G.DatumPlane(h), {"datum": h}
    elif kind == "ss":
        mdot, inv_eta, r0 = U(0.05, 0.4), U(5, 20), U(1.5, 8)
        d, od = G.ShakuraSunyaev(mdot, inv_eta, r0), {"mdot": mdot, "inv_eta": inv_eta, "inner_radius": r0}
    elif kind == "ellipse":
        e = (U(1.5, 4), U(15, 60), U(0.5, 5))
        d, od = G.EllipticalDisc(*e), {"ellipse": e}
    elif kind == "precess":
        b, g = U(0, 0.6), U(0, 6.28)
        d, od = G.PrecessingDisc(G.ThinDisc(rin, rout), b, g), {"precessing": (rin, rout, b, g)}
    elif kind == "warped":
        amp, wl = U(0.1, 1.5), U(3, 12)
        d = G.WarpedThinDisc(lambda ρ, amp=amp, wl=wl: amp * math.sin(ρ / wl), inner_radius=rin, outer_radius=rout, samples=4096)
        od = {"table": d.table, "range": d.ρ_range, "warped": True}
    else:
        r0, w, hh = U(5, 20), U(1, 5), U(0.3, 3)
        f = lambda ρ, r0=r0, w=w, hh=hh: hh * math.sqrt(max(0.0, 1 - ((ρ - r0) / w) ** 2)) if abs(ρ - r0) < w else -1.0
        d = G.ThickDisc(f, ρ_range=(max(r0 - w, 0.0), r0 + w), samples=4096)
        od = {"table": d.table, "range": d.ρ_range}
    if os.environ.get("SOAK_COMPOSITE") == "1" and kind in ("thin", "datum", "ss", "ellipse"):
        # the same scenes (same random stream) with a second geometry composed onto the first -- CompositeGeometry(d, ring):
        # an outer thin ring beyond the first geometry's extent, and for every third scene a datum plane below it as well
        ring = (rout + 5.0, rout + 60.0)
        comps_d, comps_o = [d, G.ThinDisc(*ring)], [od, ring]
        if case % 3 == 0:
            comps_d.append(G.DatumPlane(-3.0)); comps_o.append({"datum": -3.0})
        d, od, kind = G.CompositeGeometry(*comps_d), {"composite": comps_o}, kind + "+composite"
    if only is not None and case != only:
        continue
    if tol_override is not None:
        tol = tol_override
    if only is not None:
        print("scene", case, name, params, kind, od if kind not in ("table", "warped") else kind, dict(r_obs=r_obs, th=math.degrees(th), gtol=gtol,
              tol=tol, hemi=hemi, q=q, lam=lam, lim=lim))
    try:
        if HOST:
            cfgh = G.render_configuration(m, x, d, lam, image_width=W, image_height=H, alpha_lims=(-lim, lim), beta_lims=(-lim, lim),
                                          gtol=gtol, abstol=tol, reltol=tol, q=q, callback=G.domain_upper_hemisphere() if hemi else None)
            got = Hh.render_endpoints(G, cfgh)
            cache = None
        else:
          _, _, cache = G.prerendergeodesics(m, x, d, lam, image_width=W, image_height=H, alpha_lims=(-lim, lim),
                                           beta_lims=(-lim, lim), gtol=gtol, abstol=tol, reltol=tol, q=q, ensemble=ens,
                                           callback=G.domain_upper_hemisphere() if hemi else None)
          got = np.ascontiguousarray(cache.points.T).ravel()
        ocfg = oracle.make_config(name, params, disc=od, lambda_max=lam, gtol=gtol, abstol=tol, reltol=tol, upper_hemisphere=hemi, q=q)
        ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-lim, lim), (-lim, lim), W, H), nthreads=16)
    except Exception as e:      # noqa: BLE001
        bad.append((case, name, params, kind, "EXC " + str(e)[:100]))
        continue
    mism = int((got["status"] != ref["status"]).sum())
    mis += mism
    tot += got.size
    ok = (got["status"] == ref["status"]) & (ref["status"] >= 2) & (ref["flags"] == 0) & (got["flags"] == 0)
    e2 = 0.0
    if ok.any():
        scale = np.maximum(np.abs(ref["x"][ok]), 1.0)
        err = (np.abs(got["x"][ok] - ref["x"][ok]) / scale).max(axis=1)
        e2 = float(np.sort(err)[-2 if err.size > 1 else -1])
    lim_e = max(1e3 * tol, 1e-6) * (200 if name == "kerr-refractive" else 10 if name == "kerr-dark-matter" else 1)
    worst = max(worst, e2 / lim_e)
    if mism > 8 or e2 >= lim_e:
        # conditioning: which of the differing rays does the ORACLE trace differently from itself when its tolerance is
        # nudged (x0.5, x0.9, x1.1, x2)?  Either the class changes, or -- rays grazing the rim of the disc -- the class stays
        # "disc" and the crossing jumps from the rim to the far side (a position difference of order one).  Those are rays no
        # two implementations agree on; what is left is unexplained.
        flips = np.zeros(got.size, dtype=bool)
        for f in (0.5, 0.9, 1.1, 2.0):
            ocfg2 = oracle.make_config(name, params, disc=od, lambda_max=lam, gtol=gtol, abstol=f * tol, reltol=f * tol, upper_hemisphere=hemi, q=q)
            ref2 = oracle.trace(ocfg2, x, oracle.render_velocities(ocfg2, x, (-lim, lim), (-lim, lim), W, H), nthreads=16)
            flips |= ref2["status"] != ref["status"]
            flips |= (np.abs(ref2["x"] - ref["x"]) / np.maximum(np.abs(ref["x"]), 1.0)).max(axis=1) > 1e-3
        unexplained = int(((got["status"] != ref["status"]) & ~flips).sum())
        unexplained_total += unexplained
        okx = ok & ~flips
        e_un = 0.0
        if okx.any():
            e_un = float((np.abs(got["x"][okx] - ref["x"][okx]) / np.maximum(np.abs(ref["x"][okx]), 1.0)).max())
        if e_un >= lim_e:
            unexplained_err_scenes += 1
        bad.append((case, name, tuple(round(p, 4) for p in params), kind, f"mism={mism} err={e2:.2e} unexplained={unexplained} err-outside-oracle-flips={e_un:.2e} oracle-self-flips={int(flips.sum())} tol={tol} robs={r_obs:.1f} th={math.degrees(th):.1f} gtol={gtol:.4f} hemi={hemi} q={q:.2f}"))
print(f"scenes={n_scenes} seed={seed} rays={tot} status-mismatches={mis} ({mis / max(tot, 1):.4%}) worst err/limit={worst:.3f} failing={len(bad)} "
      f"mismatches in failing scenes the oracle does not flip itself under a tolerance nudge: {unexplained_total}; "
      f"failing scenes whose largest end-point difference outside the oracle's own flips exceeds the limit: {unexplained_err_scenes}")
for b in bad:
    print("  ", b)
