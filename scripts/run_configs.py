"""Runs the BASELINE.json configurations C1..C4 at full size on one MI355X and prints one line each
(C5 is scripts/run_c5_fullsize.py).  C1 is also checked against the oracle (it is the CPU-plumbing case)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)


def run(tag, m, x, d, lam, W, H, alims, blims, pf, reps=3):
    best = None
    for _ in range(reps):
        a, b, img, st = G.rendergeodesics(m, x, d, lam, image_width=W, image_height=H, alpha_lims=alims, beta_lims=blims,
                                          pf=pf, ensemble=ens, stats=True)
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    hits = int(np.isfinite(img).sum())
    print(f"{tag}: {W}x{H} rays={st['rays']} steps/ray={st['accepted_steps']/st['rays']:.1f} hits={hits} "
          f"kernel+D2H ms={best:.2f} rays/s={st['rays']/best*1e3:.3e} g range=[{np.nanmin(img):.4f}, {np.nanmax(img):.4f}] "
          f"flagged={st['flagged_rays']} status={st['status_count']}")
    return img


CPF = G.ConstPointFunctions
m = G.KerrMetric(1.0, 0.998)
x1 = np.array([0.0, 100.0, math.radians(85), 0.0])
pf = CPF.redshift(m, x1) @ CPF.filter_intersected()
img = run("C1 Kerr a=0.998 64x64 (r_obs=100)", m, x1, G.ThinDisc(0.0, 40.0), 200.0, 64, 64, (-9.5, 9.5), (-9.5, 9.5), pf)
try:
    from oracle import oracle as O
    cfg = O.make_config("kerr", (1.0, 0.998), disc=(0.0, 40.0), lambda_max=200.0)
    t0 = time.perf_counter()
    ref = O.rendergeodesics(cfg, x1, (-9.5, 9.5), (-9.5, 9.5), 64, 64, pf_id=O.PF_REDSHIFT, filter_id=O.FILTER_INTERSECTED, r_isco=m.isco())
    dt = time.perf_counter() - t0
    both = ~np.isnan(img) & ~np.isnan(ref)
    print(f"   C1 vs oracle ({O.lib().orc_max_threads()} threads, {dt*1e3:.0f} ms): class mismatches={(np.isnan(img)!=np.isnan(ref)).sum()} "
          f"max rel err={np.max(np.abs(img[both]/ref[both]-1)):.2e}")
except Exception as e:  # oracle not built
    print("   oracle unavailable:", e)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
pf = CPF.redshift(m, x) @ CPF.filter_intersected()
run("C2 Kerr a=0.998 1024x1024", m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, 1024, 1024, (-60, 60), (-35, 35), pf)
run("C3 Kerr a=0.998 2048x2048 (1 GPU)", m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, 2048, 2048, (-60, 60), (-35, 35), pf)
mj = G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0)
xj = np.array([0.0, 1000.0, math.radians(70), 0.0])
t0 = time.perf_counter()
pfj = CPF.redshift(mj, xj, ensemble=ens) @ CPF.filter_intersected()
print(f"   C4 host set-up (generic isco + device-traced plunging table): {1e3*(time.perf_counter()-t0):.0f} ms, isco={mj.isco():.6f}, table rows={pfj.extra['plunge'][0].size}")
run("C4 Johannsen(a=0.7, a13=2, e3=1) 1024x1024", mj, xj, G.ThinDisc(mj.isco(), 50.0), 2000.0, 1024, 1024, (-60, 60), (-35, 35), pfj)
