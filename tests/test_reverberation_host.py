"""lagtransfer / binflux host logic (gradus.jl_amd/reverberation.py) on oracle-traced rays, against
the exact counts and the flux sum recorded in test/transfer-functions/test-2d.jl:4-33."""
import math

import numpy as np
import pytest


def test_lagtransfer_counts_and_binned_flux(G, oracle):
    K, RV = G.corona, G.reverberation
    m = G.KerrMetric(M=1.0, a=0.998)
    x = np.array([0.0, 1e6, math.radians(30), 0.0])
    plane = G.PolarPlane(G.GeometricGrid(), Nr=20, Nθ=20)
    disc = (m.isco(), 500.0)
    model = G.LampPostModel(h=10.0, θ=math.radians(0.0001))
    max_t = 2 * x[1]
    s = G.EvenSampler(domain=G.BothHemispheres(), generator=G.GoldenSpiralGenerator())
    xs, vs, vsrc = K.sample_position_direction_velocity(m, model, s, 100)
    assert xs[0, 2] == pytest.approx(1e-3)                       # pole guard of corona-models.jl:18-24
    ccfg = oracle.make_config("kerr", (1.0, 0.998), disc=disc, lambda_max=max_t, upper_hemisphere=True)
    gps = oracle.trace(ccfg, xs, vs)
    mask = gps["status"] == oracle.INTERSECTED_WITH_GEOMETRY
    assert mask.sum() == 58                                      # length(tf.coronal_geodesics.geodesic_points)
    ce = K.CoronaGeodesics(m, G.ThinDisc(*disc), model, gps[mask], vsrc[mask])
    ocfg = oracle.make_config("kerr", (1.0, 0.998), disc=disc, lambda_max=max_t, upper_hemisphere=True,
                              outer_radius=1.1 * x[1])
    a, b = G.impact_parameters(plane, x)
    o2d = oracle.trace(ocfg, x, oracle.map_impact_parameters(ocfg, x, a, b))
    tf = RV.assemble_lagtransfer(max_t, x, plane, ce, o2d)
    assert tf.observer_to_disc.size == 337                       # length(tf.observer_to_disc)
    assert tf.image_plane_areas.size == 337
    g = oracle.apply_pf(ocfg, tf.observer_to_disc, max_t, pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_NONE,
                        r_isco=m.isco())
    t, E, f = RV.binflux(tf, g=g, N_t=100, N_E=100)
    assert f.shape == (100, 100) and t.size == 100 and E.size == 100
    assert float(np.nansum(f)) == pytest.approx(3.9126785201177956, abs=1e-2)     # the reference's tolerance
    assert float(np.nansum(f)) == pytest.approx(3.9126785201177956, rel=1e-5)     # what is reached
    # every photon lands in a cell; arrival times are delays after the direct continuum (t - t0 > 0)
    de, dt = E[1] - E[0], t[1] - t[0]
    assert float(np.nansum(f)) * de * dt == pytest.approx(1.0, rel=1e-12)
    assert t[0] > 0 and 0.05 * 6.4 < E[0] < E[-1] < 1.5 * 6.4


def test_bin_transfer_function_cells(G):
    RV = G.reverberation
    t = np.array([0.0, 0.5, 1.0, 1.0])
    e = np.array([1.0, 1.0, 2.0, 3.0])
    fl = np.array([1.0, 2.0, 3.0, 4.0])
    tb, eb, tf = RV.bin_transfer_function(t, e, fl, N_E=3, N_t=3)
    np.testing.assert_allclose(tb, [0.0, 0.5, 1.0])
    np.testing.assert_allclose(eb, [1.0, 2.0, 3.0])
    expect = np.full((3, 3), np.nan)
    expect[0, 0], expect[0, 1], expect[1, 2], expect[2, 2] = 1.0, 2.0, 3.0, 4.0
    np.testing.assert_allclose(tf, expect / 0.5)


def test_semi_analytic_lag_transfer_reference_values(G, oracle):
    """test/transfer-functions/test-2d.jl:35-79: Monte-Carlo emissivity profile (5000 golden-spiral rays)
    x transfer functions at 5 radii from the ISCO out, integrated over (g, t)."""
    K = G.corona
    m = G.KerrMetric(M=1.0, a=0.998)
    x = np.array([0.0, 1e6, math.radians(30), 0.0])
    model = G.LampPostModel(h=10.0, θ=math.radians(0.0001))
    pcfg = oracle.make_config("kerr", (1.0, 0.998), mu=1.0, closest_approach=1.000001, lambda_max=50000.0)
    proj = K.keplerian_velocity_projector(m, plunging=oracle.plunging_table(pcfg, m.isco()))
    s = G.EvenSampler(domain=G.BothHemispheres(), generator=G.GoldenSpiralGenerator())
    xs, vs, vsrc = K.sample_position_direction_velocity(m, model, s, 5000)
    ccfg = oracle.make_config("kerr", (1.0, 0.998), disc=(m.isco(), 500.0), lambda_max=10000.0, upper_hemisphere=True)
    gps = oracle.trace(ccfg, xs, vs)
    mask = gps["status"] == oracle.INTERSECTED_WITH_GEOMETRY
    prof = K.build_radial_profile(m, K.PowerLawSpectrum(2.0), gps[mask], vsrc[mask], N=100, disc_velocity=proj)
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc={"datum": 0.0}, lambda_max=2 * x[1], outer_radius=2 * x[1])

    def trace(al, be):
        pts = oracle.trace(cfg, x, oracle.map_impact_parameters(cfg, x, np.asarray(al), np.asarray(be)))
        return pts, oracle.apply_pf(cfg, pts, 2 * x[1], pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_NONE,
                                    r_isco=m.isco())

    radii = G.InverseGrid()(m.isco(), 100.0, 5)
    itb = G.transferfunctions(m, x, G.ThinDisc(0.0, 500.0), radii=radii, tracer=trace)
    bins, tbins = np.linspace(0.0, 1.5, 100), np.linspace(0.0, 150.0, 100)
    flux = G.integrate_lagtransfer(prof, itb, bins, tbins, t0=x[1], n_radii=1000, rmin=min(radii), rmax=max(radii))
    assert flux.shape == (100, 100)
    assert float(flux.sum()) == pytest.approx(1.0, abs=1e-2)
    assert float(flux[39, :].sum()) == pytest.approx(0.021759503160585468, abs=1e-4)      # flux[40, :] in Julia
    assert float(flux[39, :].sum()) == pytest.approx(0.021759503160585468, rel=2e-3)
    # the response starts after the light-travel delay and the red wing arrives first from small radii
    first = np.argmax(flux.sum(axis=0) > 0)
    assert tbins[first] > 5.0
    # lag-frequency spectrum of this response: finite, positive lag at low frequencies
    freq, tau = G.lag_frequency(tbins, flux)
    assert freq.size == tau.size and freq[1] == pytest.approx(5e-5, rel=0.02)
    assert np.all(np.isfinite(tau[1:50])) and tau[1] > 0


def test_lag_frequency_of_a_delayed_pulse(G):
    """a δ-response at delay T with small amplitude has lag ≈ A sin(2πνT)/(2πν(1 + A cos(2πνT))) -> A·T at ν -> 0"""
    t = np.arange(0.0, 400.0, 1.0)
    psi = np.zeros_like(t)
    psi[50] = 0.1
    freq, tau = G.lag_frequency(t, psi)
    nu = freq[1:20]
    expect = np.arctan(0.1 * np.sin(2 * np.pi * nu * 50.0) / (1 + 0.1 * np.cos(2 * np.pi * nu * 50.0))) / (2 * np.pi * nu)
    np.testing.assert_allclose(tau[1:20], expect, rtol=1e-9, atol=1e-12)


def test_reverberation_smoke_chain_reference_values(G, oracle):
    """test/smoke-tests/reverberation.jl:1-45: continuum time, angular emissivity profile (500 rays),
    transfer functions at 10 radii with the image-plane origin offset β₀ = 2, (g, t) integration and
    the lag-frequency spectrum: sum(freq) and τ[132] as recorded (rtol 1e-2 there)."""
    K, RV = G.corona, G.reverberation
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 10_000.0, math.radians(45), 0.0])
    model = G.LampPostModel()
    pos, vsrc = model.sample_position_velocity(m)
    pcfg = oracle.make_config("kerr", (1.0, 0.998), disc={"datum": pos[1] * math.cos(pos[2])}, lambda_max=2 * x[1],
                              outer_radius=2 * x[1])
    t0 = RV.continuum_time(m, x, model, tracer=lambda a, b: oracle.trace(
        pcfg, x, oracle.map_impact_parameters(pcfg, x, np.asarray(a), np.asarray(b))))
    # light travel time from r = 5 on the axis to r = 1e4 at 45°: a little more than the coordinate distance
    assert 10_000.0 < t0 < 10_030.0
    plcfg = oracle.make_config("kerr", (1.0, 0.998), mu=1.0, closest_approach=1.000001, lambda_max=50000.0)
    proj = K.keplerian_velocity_projector(m, plunging=oracle.plunging_table(plcfg, m.isco()))
    ds = np.radians(np.linspace(0.01, 179.99, 500))
    ccfg = oracle.make_config("kerr", (1.0, 0.998), disc=(0.0, float("inf")), lambda_max=10000.0, upper_hemisphere=True)
    gps = oracle.trace(ccfg, pos, K.polar_angle_velocities(m, pos, vsrc, ds))
    prof = K.point_source_profile_from_points(m, K.PowerLawSpectrum(2.0), vsrc, ds, gps, proj)
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc={"datum": 0.0}, lambda_max=2 * x[1], outer_radius=2 * x[1])

    def trace(al, be):
        pts = oracle.trace(cfg, x, oracle.map_impact_parameters(cfg, x, np.asarray(al), np.asarray(be)))
        return pts, oracle.apply_pf(cfg, pts, 2 * x[1], pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_NONE,
                                    r_isco=m.isco())

    radii = G.InverseGrid()(m.isco(), 100.0, 10)
    itb = G.transferfunctions(m, x, G.ThinDisc(0.0, float("inf")), radii=radii, tracer=trace, β0=2.0)
    bins, tbins = np.linspace(0.0, 1.5, 100), np.linspace(0.0, 100.0, 100)
    flux = G.integrate_lagtransfer(prof, itb, bins, tbins, t0=t0, n_radii=100, h=1e-8, rmin=min(radii), rmax=max(radii))
    flux[flux == 0] = np.nan
    freq, tau = G.lag_frequency(tbins, flux)
    assert float(freq.sum()) == pytest.approx(2449.8787687490535, rel=1e-2)
    assert float(tau[131]) == pytest.approx(9.322742661315855, rel=1e-2)          # τ1[132] in Julia
    assert float(freq.sum()) == pytest.approx(2449.8787687490535, rel=1e-12)
