"""A selection of the reference's `docs/src/examples.md` -- shadow, redshift image, line profile,
reverberation (binned and semi-analytic), interpolated redshift, disc geometries, circular orbits,
ISCO, horizons, transfer functions -- with `ensemble = EnsembleMI355X()`.  Plots are replaced by
printed summaries.   python examples/examples.py   (needs an MI355X)"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
CPF = G.ConstPointFunctions


def timed(f):
    t = time.perf_counter()
    r = f()
    return r, time.perf_counter() - t


# ## Tracing geodesic paths
m = G.JohannsenPsaltisMetric(M=1.0, a=0.6, eps3=2.0)
x = np.array([0.0, 10000.0, math.pi / 2, 0.0])
α = np.linspace(-10.0, 10.0, 20)
vs = G.map_impact_parameters(m, x, α, np.zeros_like(α))
sols = G.tracegeodesic_paths(m, np.tile(x, (20, 1)), vs, 20000.0, ensemble=ens)
print(f"paths: 20 Johannsen-Psaltis geodesics, {sum(p.λ.size for p in sols)} saved steps")
sols = G.corona.tracegeodesics(G.KerrMetric(a=0.0), G.LampPostModel(), 2000.0, n_samples=64, ensemble=ens)
print(f"paths: 64 lamp-post rays, statuses {np.bincount(sols['status'], minlength=4).tolist()}")

# ## Shadow
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 10000.0, math.pi / 2, 0.0])
(_, _, img), dt = timed(lambda: G.rendergeodesics(m, x, 20_000.0, image_width=800, image_height=800, alpha_lims=(-4, 8),
                                                  beta_lims=(-6, 6), ensemble=ens))
print(f"shadow: 800² in {dt:.2f} s, {np.isfinite(img).sum()} pixels inside")

# ## Redshift image (+ histogram line profile)
m = G.KerrMetric(M=1.0, a=1.0)
x = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(1.0, 50.0)
pf = CPF.redshift(m, x) @ CPF.filter_intersected()
_, _, img = G.rendergeodesics(m, x, d, 2000.0, alpha_lims=(-60, 60), beta_lims=(-30, 35), image_width=800, image_height=400,
                              pf=pf, ensemble=ens)
hist, _ = np.histogram(img[np.isfinite(img)] * 6.4, bins=np.linspace(0.0, 10.0, 100))
print(f"redshift image: g in [{np.nanmin(img):.3f}, {np.nanmax(img):.3f}]; iron-line histogram peaks at {np.linspace(0, 10, 100)[np.argmax(hist)]:.2f} keV")

# ## Line profiles (default method: integrated Cunningham transfer functions)
d = G.ThinDisc(0.0, 400.0)
x = np.array([0.0, 1000.0, math.radians(40), 0.0])
m = G.KerrMetric(1.0, 0.998)
gs = np.linspace(0.0, 1.2, 500)
(_, flux), dt = timed(lambda: G.lineprofile(gs, lambda r: r ** -3.0, m, x, d, maxrₑ=50.0, ensemble=ens))
print(f"line profile (100 transfer functions + integration): {dt:.2f} s, peak at g = {gs[int(np.argmax(flux))]:.3f}")

# ## Reverberation transfer functions: binning ...
x = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(0.0, 1000.0)
model = G.LampPostModel(h=10.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=1800, Nθ=1800)
tf, dt = timed(lambda: G.lagtransfer(m, x, d, model, plane=plane, n_samples=100_000, ensemble=ens,
                                     sampler=G.EvenSampler(G.BothHemispheres(), G.RandomGenerator(seed=1))))
t0 = G.continuum_time(m, x, model, ensemble=ens)
t, E, f = G.binflux(tf, N_E=1500, N_t=1500, t0=t0, ensemble=ens)
print(f"lagtransfer: 3.24e6 + 1e5 rays in {dt:.2f} s; continuum time {t0:.2f}; response from t = {t[np.nanargmax(np.nansum(f, axis=0) > 0)]:.1f}")
# ... and the semi-analytic route
d = G.ThinDisc(0.0, float("inf"))
radii = G.InverseGrid()(m.isco(), 1000.0, 100)
itb, dt = timed(lambda: G.transferfunctions(m, x, d, radii=radii, ensemble=ens))
prof = G.emissivity_profile(m, d, model, n_samples=2000, ensemble=ens)
gbins, tbins = np.linspace(0.0, 1.4, 500), np.linspace(0.0, 150.0, 500)
flux = G.integrate_lagtransfer(prof, itb, gbins, tbins, t0=t0, n_radii=6000)
freq, tau = G.lag_frequency(tbins, flux)
print(f"semi-analytic: 100 transfer functions in {dt:.2f} s; lag at {freq[5]:.2e} Hz·GM/c³ = {tau[5]:.2f}")

# ## Interpolating redshifts
m = G.KerrMetric(M=1.0, a=0.4)
x = np.array([0.0, 1000.0, math.radians(85), 0.0])
d = G.ThinDisc(1.0, 50.0)
pl_int = G.interpolate_plunging_velocities(m, ensemble=ens)
pf = G.interpolate_redshift(pl_int, x) @ CPF.filter_intersected()
_, _, img = G.rendergeodesics(m, x, d, 2000.0, image_width=700, image_height=240, pf=pf, ensemble=ens)
_, _, ref = G.rendergeodesics(m, x, d, 2000.0, image_width=700, image_height=240, ensemble=ens,
                              pf=CPF.redshift(m, x) @ CPF.filter_intersected())
both = np.isfinite(img) & np.isfinite(ref)
print(f"interpolated vs analytic Kerr redshift: max rel diff {np.max(np.abs(img[both] / ref[both] - 1)):.1e} on {both.sum()} pixels")

# ## Disc geometries
m = G.KerrMetric(1.0, 0.2)
x = np.array([0.0, 1000.0, math.radians(80), 0.0])
for name, d in (("ThickDisc(ρ -> …)", G.ThickDisc(lambda ρ: 0.0 if ρ < 9 or ρ > 11 else math.sqrt(1 - (ρ - 10) ** 2), ρ_range=(9.0, 11.0))),
                ("ShakuraSunyaev", G.ShakuraSunyaev.for_metric(m, eddington_ratio=0.3)),
                ("PolishDoughnut", G.PolishDoughnut(m, rₖ=12.0, n=0.21))):
    _, _, img = G.rendergeodesics(m, x, d, 2000.0, image_width=600, image_height=300, alpha_lims=(-30, 30), beta_lims=(-15, 15),
                                  pf=CPF.redshift(m, x) @ CPF.filter_intersected(), ensemble=ens)
    print(f"geometry {name}: {np.isfinite(img).sum()} pixels, g in [{np.nanmin(img):.3f}, {np.nanmax(img):.3f}]")

# ## Circular orbits, ISCO
m = G.KerrMetric(M=1.0, a=0.8)
for r in (3.0, 6.0):
    v = G.CircularOrbits.fourvelocity(m, r)
    path = G.tracegeodesic_path(m, np.array([0.0, r, math.pi / 2, 0.0]), v, (0.0, 300.0), μ=1.0, ensemble=ens)
    print(f"circular orbit at r = {r}: radius stays within {np.ptp(path.x[:, 1]):.1e} over {path.x[-1, 3] / (2 * math.pi):.1f} turns")
print("ISCO and its energy:", [(round(G.KerrMetric(1.0, a).isco(), 4), round(float(G.CircularOrbits.energy(G.KerrMetric(1.0, a), G.KerrMetric(1.0, a).isco())), 5))
                               for a in (0.0, 0.4, 0.6)])

# ## Event horizons and naked singularities
for a in (0.0, 0.5, 0.8):
    mj = G.JohannsenPsaltisMetric(M=1.0, a=a, eps3=2.0)
    rs, θs = G.event_horizon(mj, resolution=200)
    print(f"horizon of Johannsen-Psaltis(a = {a}, ϵ3 = 2): r in [{np.nanmin(rs):.3f}, {np.nanmax(rs):.3f}], naked: {G.is_naked_singularity(mj)}")

# ## Cunningham transfer functions
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 10_000.0, math.radians(75), 0.0])
ctfs, dt = timed(lambda: G.cunningham_transfer_functions(m, x, G.ThinDisc(0.0, float("inf")), [4.0, 7.0, 10.0, 12.0], ensemble=ens))
print(f"transfer functions at rₑ = 4, 7, 10, 12 in {dt:.2f} s: g ranges " + ", ".join(f"[{c.gmin:.3f}, {c.gmax:.3f}]" for c in ctfs))

# ## Photon rings: the order of every image-plane ray (TraceWindings)
m = G.KerrMetric(1.0, 0.9)
x = np.array([0.0, 1000.0, math.radians(80), 0.0])
(_, _, img), dt = timed(lambda: G.rendergeodesics(m, x, 2000.0, image_width=1024, image_height=1024, alpha_lims=(-8, 8), beta_lims=(-8, 8),
                                                  pf=CPF.winding(), trace=G.TraceWindings(), ensemble=ens))
print(f"photon rings: 1024² winding numbers in {dt:.2f} s; pixels of order 0..4+: {[int((img == k).sum()) for k in range(4)] + [int((img >= 4).sum())]}")

# ## Transfer functions of a thick disc, the ring of an emission radius as the observer sees it, a ray through a point
ss = G.ShakuraSunyaev.for_metric(m, eddington_ratio=0.3)
xo = np.array([0.0, 10_000.0, math.radians(75), 0.0])
ctf, dt = timed(lambda: G.cunningham_transfer_function(m, xo, ss, 4.0, β0=2.0, ensemble=ens))
print(f"thick-disc transfer function at rₑ = 4: {np.isfinite(ctf.f).sum()} of {ctf.f.size} samples visible, g in [{ctf.gmin:.3f}, {ctf.gmax:.3f}] ({dt:.2f} s)")
a_, b_ = G.impact_parameters_for_radius_obscured(m, x, ss, 4.0, N=200, β0=2.0, ensemble=ens)
print(f"ring at rₑ = 4 behind a Shakura-Sunyaev disc at 80°: {np.isfinite(a_).sum()} of 200 directions visible")
α_, β_, acc = G.impact_parameters_for_target(np.array([10.0, math.radians(40), -math.pi / 4]), m, np.array([0.0, 1000.0, math.pi / 2, 0.0]), ensemble=ens)
print(f"ray through (r, θ, ϕ) = (10, 40°, -45°): α = {α_:.4f}, β = {β_:.4f}, passes within {acc:.1e}")

# ## More geometry: a warped sheet and a tilted disc
for name, d in (("WarpedThinDisc", G.WarpedThinDisc(lambda ρ: 1.5 * math.sin(ρ / 5.0), inner_radius=m.isco(), outer_radius=40.0)),
                ("PrecessingDisc", G.PrecessingDisc(G.ThinDisc(m.isco(), 40.0), 0.4, 0.7))):
    _, _, img = G.rendergeodesics(m, x, d, 2000.0, image_width=600, image_height=400, alpha_lims=(-45, 45), beta_lims=(-30, 30),
                                  pf=CPF.redshift(m, x) @ CPF.filter_intersected(), ensemble=ens)
    print(f"geometry {name}: {np.isfinite(img).sum()} pixels, g in [{np.nanmin(img):.3f}, {np.nanmax(img):.3f}]")
