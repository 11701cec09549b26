#!/usr/bin/env python3
"""Soak test of GR_METRIC_TABULATED through the fp32 kernels ("precision" 32): N random scenes (the families, observers and discs of
scripts/soak_tab.py), each traced three times -- the table through the fp32 kernels, the same metric's OWN fp32 kernels, and the
fused fp64 kernels at 1e-9 as the reference.  Single precision at a tolerance of 1e-5 ... 1e-6 is coarse whatever evaluates the
metric, so the table is held to what the fused fp32 kernels do:

    per scene   status flips against fp64:  tab32 <= 1.3 x max(fused32, fused64 at the same tolerance) + 5 % of the rays (+ 3)
                rays without a status:      tab32 <= max(fused32, fused64 at the same tolerance) + 5 % (+ 3)
                median |Δx| / |x| of hits:  tab32 <= 2 x fused32 + 1e-6;  no NaN end points
    over all    flips and rays without a status of tab32 within 2 % of fused32's totals

(At these tolerances a step is longer than the disc's slab is thick and many rays step over it -- they end without a status; WHICH
rays do is chaotic in the placement of the steps: the same scene gives 1337 / 1386 hits through table / fused kernels at 7.7e-6 and
2080 / 2072 at 1e-6, fp64 1462 and 2068 -- scripts/debug_tab_fp32c.py.  Over 200 scenes the table is above the fused kernels in 99 and
below in 87; the per-scene allowance is that scatter, the totals carry the comparison.)

    python scripts/soak_tab32.py [n_scenes] [seed]
"""
import math, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradus_jl_amd as G
warnings.simplefilter("ignore")
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
ens = G.EnsembleMI355X(0)
bad, done = [], 0
tot = np.zeros(6, dtype=np.int64)      # flips tab32 / fused32 / fused64lo, no status tab32 / fused32 / fused64lo
t0 = time.time()
for case in range(n_scenes):
    rng = np.random.default_rng([seed, case])
    U = lambda a, b: float(rng.uniform(a, b))
    fam = [lambda: G.KerrMetric(1.0, U(-0.998, 0.998)), lambda: G.JohannsenMetric(1.0, U(0, 0.9), U(-1, 2), U(-1, 1), U(-1, 1), U(-1, 2)),
           lambda: (lambda a: G.KerrNewmanMetric(1.0, a, U(0, math.sqrt(1 - a * a) * 0.95)))(U(0, 0.9)),
           lambda: G.JohannsenPsaltisMetric(1.0, U(0, 0.8), U(-0.5, 1)), lambda: G.DilatonAxion(1.0, U(0.1, 0.8), U(-0.3, 0.3), U(0.3, 1.5)),
           lambda: G.KerrDarkMatter(1.0, U(0, 0.9), U(0, 3), U(5, 30), U(5, 20)), lambda: G.KerrRefractive(1.0, U(0, 0.9), U(0.9, 1.3), U(10, 30))]
    base = fam[int(rng.integers(0, len(fam)))]()
    r_obs, th = float(10 ** U(1.5, 3.2)), float(np.radians(U(8, 88)))
    rin = U(0, 8)
    disc = G.ThinDisc(rin, rin + 10 ** U(0.5, 2.3))
    kernel, tol = int(rng.integers(0, 3)), float(10 ** U(-6, -5))
    w, h, fov = int(rng.integers(8, 97)), int(rng.integers(8, 97)), U(4, 30)
    if only is not None and case != only:
        continue
    x = np.array([0.0, r_obs, th, 0.0])
    desc = f"{case}: {base} r_obs={r_obs:.1f} th={math.degrees(th):.1f} {disc} {w}x{h} fov={fov:.1f} kernel={kernel} tol={tol:.1e}"
    try:
        tab = G.TabulatedMetric(base, r_max=max(12000.0, 3 * r_obs), max_refinements=1)
        chart = G.chart_for_metric(tab, 2 * r_obs)
        run = lambda m, prec, t: (ens.set("kernel", kernel).set("precision", prec),
                                  G.prerendergeodesics(m, x, disc, 2 * r_obs, image_width=w, image_height=h, alpha_lims=(-fov, fov),
                                                       beta_lims=(-fov, fov), abstol=t, reltol=t, chart=chart, ensemble=ens)[2].points.ravel())[1]
        ref, f32, t32, f64lo = run(base, 64, 1e-9), run(base, 32, tol), run(tab, 32, tol), run(base, 64, tol)
    except Exception as e:
        print("SKIP", desc, "->", type(e).__name__, str(e)[:100]); continue
    finally:
        ens.set("precision", 64).set("kernel", 2)
    n = ref.size
    def against(p):
        flips = int((p["status"] != ref["status"]).sum())
        hit = (p["status"] == 2) & (ref["status"] == 2)
        err = np.abs(p["x"][hit, 1:3] - ref["x"][hit, 1:3]).max(axis=1) / np.maximum(np.abs(ref["x"][hit, 1]), 1.0) if hit.any() else np.zeros(1)
        return flips, float(np.median(err)), int((p["status"] == 3).sum()), bool(np.all(np.isfinite(p["x"])))
    ft, et, st, fin_t = against(t32)
    ff, ef, sf, fin_f = against(f32)
    fl, el, sl, _ = against(f64lo)
    ok = ft <= 1.3 * max(ff, fl) + 0.05 * n + 3 and et <= 2 * ef + 1e-6 and st <= max(sf, sl) + 0.05 * n + 3 and fin_t
    done += 1
    tot += np.array([ft, ff, fl, st, sf, sl])
    if os.environ.get("SOAK_VERBOSE"):
        d = np.nonzero((t32["status"] == 3) != (f32["status"] == 3))[0]
        print("  rays stuck in one of the fp32 traces only:", d.size)
        for i in d[:12]:
            print(f"   ray {i}: ref st {ref['status'][i]} r {ref['x'][i, 1]:.4f} | fused32 st {f32['status'][i]} flags {f32['flags'][i]:#x} r {f32['x'][i, 1]:.4f} th {f32['x'][i, 2]:.4f} lam {f32['lambda_max'][i]:.3f}"
                  f" | tab32 st {t32['status'][i]} flags {t32['flags'][i]:#x} r {t32['x'][i, 1]:.4f} th {t32['x'][i, 2]:.4f} lam {t32['lambda_max'][i]:.3f}")
        print("  inner radius", base.inner_radius(), "table r_min", tab.r_min, "chart inner", chart.inner_radius if hasattr(chart, "inner_radius") else chart)
    if not ok:
        bad.append(case)
    print("ok  " if ok else "FAIL", desc, f"rays={n} flips tab32/fused32/fused64lo {ft}/{ff}/{fl} median err {et:.1e}/{ef:.1e} no-status {st}/{sf}/{sl}", flush=True)
agg = only is not None or (tot[0] <= 1.02 * tot[1] + 3 and tot[3] <= 1.02 * tot[4] + 3)
print(f"\n{n_scenes} scenes ({done} compared), seed {seed}: flips tab32 / fused32 / fused64 at the same tolerance {tot[0]} / {tot[1]} / {tot[2]}, "
      f"without status {tot[3]} / {tot[4]} / {tot[5]}: totals {'ok' if agg else 'FAIL'}; failing scenes: {bad}   [{time.time() - t0:.0f} s]")
sys.exit(1 if bad or not agg else 0)
