"""The reference's "Getting started" walk-through (docs/src/getting-started.md §2-8) with
`ensemble = EnsembleMI355X()`: same calls, same names (Julia's `∘` is written `@`), every geodesic
traced on the MI355X.  Run on a machine with an MI355X:  python examples/getting_started.py

Differences from the Julia text: a metric is chosen from the device's catalogue instead of being
defined by a `metric_components` closure (user closures cannot cross the C ABI -- with such a metric
keep a CPU ensemble), and plots are replaced by printed summaries."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gradus_jl_amd as G

ens = G.EnsembleMI355X(0)
CPF = G.ConstPointFunctions

# -- 2. photon trajectories -----------------------------------------------------------------
m = G.KerrMetric(M=1.0, a=0.0)                                   # the text's Schwarzschild(1.0)
x = np.array([0.0, 1000.0, math.pi / 2, 0.0])
v = np.array([0.0, -1.0, 0.0, -8e-6])
λ_max = 2000.0
sol = G.tracegeodesic_path(m, x, v, λ_max, ensemble=ens)         # every accepted step saved
print(f"§2  one trajectory: {sol.λ.size} steps, closest approach r = {sol.x[:, 1].min():.3f}, status {int(sol.point['status'])}")

α = np.linspace(-10.0, 10.0, 30)
vs = G.map_impact_parameters(m, x, α, np.zeros_like(α))
xs = np.tile(x, (vs.shape[0], 1))
sols = G.tracegeodesic_paths(m, xs, vs, λ_max, ensemble=ens)
print(f"§2  30 trajectories: {sum(p.λ.size for p in sols)} saved steps, "
      f"{sum(int(p.point['status']) == G.StatusCodes.WithinInnerBoundary for p in sols)} captured")

# -- 3. a basic shadow ------------------------------------------------------------------------
α = β = np.linspace(-10.0, 10.0, 100)
A, B = np.meshgrid(α, β, indexing="ij")
vs = G.map_impact_parameters(m, x, A.ravel(), B.ravel())
points = G.tracegeodesics(m, x, vs, λ_max, ensemble=ens).reshape(100, 100)      # end points only (save_on = false)
times = np.where(points["status"] == G.StatusCodes.WithinInnerBoundary, points["x"][..., 0], np.nan)
print(f"§3  shadow: {np.isfinite(times).sum()} of 10000 rays captured (critical impact parameter 3√3 -> "
      f"{math.pi * 27 / 0.2 ** 2 / 1e4 * 1e4:.0f} expected)")

# -- 4. point functions -------------------------------------------------------------------------
time_coord = G.PointFunction(lambda m_, gp, λ: gp["x"][0])
filter_event_horizon = G.FilterStatusCode(G.StatusCodes.WithinInnerBoundary)
pf = time_coord @ filter_event_horizon                            # Julia: time_coord ∘ filter_event_horizon
t0 = time.perf_counter()
a_ax, b_ax, image = G.rendergeodesics(m, x, λ_max, pf=pf, image_width=800, image_height=800, alpha_lims=(-10, 10),
                                      beta_lims=(-8, 8), ensemble=ens)
print(f"§4  800x800 render with a custom Python point function on device-traced end points: "
      f"{time.perf_counter() - t0:.2f} s, {np.isfinite(image).sum()} shadow pixels")

# -- 5. adding geometry -------------------------------------------------------------------------
def cross_section(ρ):
    center, radius = 8.0, 3.0
    if ρ < center - radius or radius + center < ρ:
        return 0.0
    r = ρ - center
    return math.sqrt(radius ** 2 - r ** 2) + 0.5 * math.sin(3 * ρ)


d = G.ThickDisc(cross_section, ρ_range=(4.0, 12.0))               # the closure is sampled for the device
pf_geometry = time_coord @ CPF.filter_intersected()
x = np.array([0.0, 1000.0, math.radians(70), 0.0])
kw = dict(image_width=1200, image_height=800, alpha_lims=(-20, 20), beta_lims=(-15, 15), ensemble=ens)
_, _, image = G.rendergeodesics(m, x, d, λ_max, pf=pf_geometry, **kw)
print(f"§5  torus with a wavy cross-section: {np.isfinite(image).sum()} of {image.size} pixels hit it")

# -- 6. physical quantities ----------------------------------------------------------------------
redshift_geometry = CPF.redshift(m, x) @ CPF.filter_intersected()     # fused into the trace kernel
_, _, image = G.rendergeodesics(m, x, d, λ_max, pf=redshift_geometry, **kw)
print(f"§6  redshift image: g in [{np.nanmin(image):.3f}, {np.nanmax(image):.3f}]")

# -- 7. changing metric ---------------------------------------------------------------------------
j_m = G.JohannsenMetric(M=1.0, a=0.7, alpha13=2.0, eps3=1.0)
j_redshift_geometry = CPF.redshift(j_m, x, ensemble=ens) @ CPF.filter_intersected()
_, _, image = G.rendergeodesics(j_m, x, d, λ_max, pf=j_redshift_geometry, **kw)
print(f"§7  Johannsen(a = 0.7, α13 = 2, ϵ3 = 1): g in [{np.nanmin(image):.3f}, {np.nanmax(image):.3f}], isco {j_m.isco():.4f}")

# -- 8. line profiles --------------------------------------------------------------------------------
bins = np.linspace(0.1, 1.4, 200)
plane = G.PolarPlane(G.GeometricGrid(), Nr=1000, Nθ=1000, r_max=50.0)


def calculate_line_profile(m_, x_, d_):
    t = time.perf_counter()
    _, f = G.lineprofile(bins, G.PowerLawEmissivity(3), m_, x_, d_, G.BinningMethod(), plane=plane, maxrₑ=50.0,
                         callback=G.domain_upper_hemisphere(), ensemble=ens)
    return f, time.perf_counter() - t


d_thin = G.ThinDisc(0.0, 1000.0)
for name, mm in (("Johannsen", j_m), ("Schwarzschild", m)):
    f, dt = calculate_line_profile(mm, x, d_thin)
    print(f"§8  line profile, {name}: 10⁶ rays in {dt:.2f} s (the text: ~30 s on a 2021 M1 laptop); "
          f"peak at g = {bins[int(np.argmax(f))]:.3f}, red edge {bins[int(np.argmax(f > 0))]:.3f}")

# -- beyond the walk-through: coronae, transfer functions, lags --------------------------------------
mk = G.KerrMetric(1.0, 0.998)
model = G.LampPostModel(h=10.0)
prof = G.emissivity_profile(mk, G.ThinDisc(0.0, 500.0), model, n_samples=2000, ensemble=ens)
i6 = int(np.argmin(np.abs(prof.radii - 6.0)))
print(f"    lamp-post emissivity (h = 10): ε(r ≈ 6) / ε(r ≈ 60) = "
      f"{prof.emissivity_at(6.0) / prof.emissivity_at(60.0):.1f} over {prof.radii.size} disc hits")
xo = np.array([0.0, 10_000.0, math.radians(45), 0.0])
t = time.perf_counter()
tfs = G.transferfunctions(mk, xo, G.ThinDisc(0.0, float("inf")), numrₑ=60, ensemble=ens)
flux = G.integrate_lineprofile(lambda r: r ** -3.0, tfs, bins)
print(f"    60 Cunningham transfer functions + integrated line profile: {time.perf_counter() - t:.2f} s, "
      f"blue horn at g = {bins[int(np.argmax(flux))]:.3f}")

# -- round 5: the Monte-Carlo emissivity on the device, and a metric of the user's own ----------------------
t = time.perf_counter()
mc = G.emissivity_profile(mk, G.ThinDisc(0.0, 500.0), model, n_samples=1_000_000, N=100, ensemble=ens,
                          sampler=G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator()))
print(f"    Monte-Carlo emissivity, 10⁶ sky samples formed, traced, reduced and binned on the device: "
      f"{time.perf_counter() - t:.3f} s, ε(6) / ε(60) = {mc.emissivity_at(6.0) / mc.emissivity_at(60.0):.1f}")


def quadrupole_kerr(r, θ, a=0.9, ϵ=0.15):
    """A metric no catalogue holds, written the way a user of the reference writes `metric_components`: Kerr with
    g_tt and g_rr deformed by ϵ M³ / r³ (numpy arrays in, arrays out)."""
    s2, c2 = np.sin(θ) ** 2, np.cos(θ) ** 2
    Σ, Δ = r * r + a * a * c2, r * r - 2.0 * r + a * a
    h = ϵ / r ** 3
    A = (r * r + a * a) ** 2 - a * a * Δ * s2
    return (-(1.0 - 2.0 * r / Σ) * (1.0 + h), Σ / Δ * (1.0 + h), Σ, A * s2 / Σ, -2.0 * a * r * s2 / Σ)


own = G.TabulatedMetric(quadrupole_kerr, inner_radius=1.0 + math.sqrt(1.0 - 0.81), r_max=3000.0)
t = time.perf_counter()
pf_own = G.ConstPointFunctions.redshift(own, x, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
_, _, img_own = G.rendergeodesics(own, x, G.ThinDisc(own.isco(), 50.0), 2000.0, image_width=1024, image_height=1024,
                                  alpha_lims=(-60, 60), beta_lims=(-35, 35), pf=pf_own, ensemble=ens)
print(f"    a user-defined metric through the table ({own.table.nbytes / 1e6:.1f} MB, fit estimates {own.errors[0]:.0e} / "
      f"{own.errors[1]:.0e}): 1024² redshift image in {time.perf_counter() - t:.2f} s, isco {own.isco():.4f}, "
      f"g in [{np.nanmin(img_own):.3f}, {np.nanmax(img_own):.3f}]")
