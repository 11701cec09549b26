"""Corona -> disc on the device (gr_rayset.sky_*, gr_corona_trace, gr_corona_bin) against the record route:
`tracegeodesics(m, xs, vs, d, λ)` on host-built (x, v) arrays + numpy's reductions (corona.build_radial_profile), which is
itself pinned to the oracle and to the reference's golden vectors in test_corona_host.py / test_gpu_parity.py.

Reference: src/corona/{samplers,corona-models,emissivity,radial,flux-calculations}.jl.
"""
import ctypes as C
import math
import time

import numpy as np
import pytest


@pytest.fixture(scope="module")
def G(built):
    import gradus_jl_amd as G

    return G


@pytest.fixture(scope="module")
def ens(G):
    return G.EnsembleMI355X(device=0)


def _models(G):
    kerr = G.KerrMetric(1.0, 0.998)
    return [
        ("lamp-post / Kerr", kerr, G.LampPostModel(h=10.0), G.ThinDisc(0.0, 500.0)),
        ("beamed point source / Johannsen", G.JohannsenMetric(M=1.0, a=0.6, alpha13=0.5, eps3=0.3), G.BeamedPointSource(8.0, 0.3),
         G.ThinDisc(0.0, 400.0)),
        ("co-rotating ring / Kerr", G.KerrMetric(1.0, 0.9), G.RingCorona(G.SourceVelocities.co_rotating, 8.0, 3.0),
         G.ThinDisc(0.0, 200.0)),
    ]


def _samplers(G):
    out = []
    for Sampler in (G.EvenSampler, G.WeierstrassSampler):
        for gen in (G.GoldenSpiralGenerator, G.EvenGenerator, lambda: G.RandomGenerator(seed=11)):
            for dom in (G.LowerHemisphere, G.BothHemispheres):
                out.append(lambda Sampler=Sampler, gen=gen, dom=dom: Sampler(domain=dom(), generator=gen()))
    return out


# ---------------- not gpu: the host half ----------------
def test_sky_rayset_matrix_reproduces_sky_angles_to_velocity(G):
    """v = Mx (1, k̂) with Mx = T diag(1, J) is samplers.jl:81-99 for every source in the list."""
    K = G.corona
    for _, m, model, _d in _models(G):
        s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
        rs, keep, x, v = K.sky_rayset(m, model, s, 64)
        Mx = np.array(list(rs.Mx)).reshape(4, 4)
        xs, vs, vsrc = K.sample_position_direction_velocity(m, model, s, 64)
        i = K.geti(s, np.arange(1, 65), 64)
        θ, ϕ = K.sample_angles(s, i, 64)
        khat = -np.stack([np.sin(θ) * np.cos(ϕ), np.sin(θ) * np.sin(ϕ), np.cos(θ)], axis=-1)
        got = np.concatenate([np.ones((64, 1)), khat], axis=1) @ Mx.T
        np.testing.assert_allclose(got, vs, rtol=0, atol=1e-14)
        np.testing.assert_array_equal(np.array(list(rs.x_obs)), xs[0])
        np.testing.assert_array_equal(v, vsrc[0])
        assert (rs.sky_sampler, rs.sky_both, rs.sky_generator, rs.n) == (1, 1, 0, 64) and keep is None


def test_sky_rayset_of_a_source_without_one_position_brings_rows(G):
    """DiscCorona (extended.jl:165-183): every sample leaves from its own point of the disc -- 28 doubles per sample cross the
    boundary (gr_rayset.sky_rows): position, the matrix T diag(1, J) there, the source's velocity with its index lowered, g_tμ."""
    K = G.corona
    m = G.KerrMetric(1.0, 0.5)
    model, twin = G.DiscCorona(G.SourceVelocities.co_rotating, 8.0, 5.0, seed=3), G.DiscCorona(G.SourceVelocities.co_rotating, 8.0, 5.0, seed=3)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    rs, keep, x, v = K.sky_rayset(m, model, s, 32)
    rows = keep[1]
    assert rs.sky_rows == rows.ctypes.data and rows.shape == (32, 28) and v.shape == (32, 4)
    xs, vs, vsrc = K.sample_position_direction_velocity(m, twin, s, 32)          # the same draws, the host's way
    np.testing.assert_array_equal(rows[:, 0:4], xs)
    np.testing.assert_array_equal(v, vsrc)
    assert np.ptp(xs[:, 1]) > 1.0                                                   # the samples do sit at different places
    i = K.geti(s, np.arange(1, 33), 32)
    θ, ϕ = K.sample_angles(s, i, 32)
    khat = -np.stack([np.sin(θ) * np.cos(ϕ), np.sin(θ) * np.sin(ϕ), np.cos(θ)], axis=-1)
    got = np.einsum("kij,kj->ki", rows[:, 4:20].reshape(32, 4, 4), np.concatenate([np.ones((32, 1)), khat], axis=1))
    np.testing.assert_allclose(got, vs, rtol=0, atol=1e-13)
    # the lowered source velocity and g_tμ: (u · v) / (g_tμ v^μ) is what turns the static observer's ratio into the source's
    g = [m.metric_components(r, th) for r, th in xs[:, 1:3]]
    for k in range(32):
        e_src = K._dot(g[k], vs[k], vsrc[k])
        assert rows[k, 20:24] @ vs[k] == pytest.approx(e_src, rel=1e-13)
        assert rows[k, 24:28] @ vs[k] == pytest.approx(g[k][0] * vs[k, 0] + g[k][4] * vs[k, 3], rel=1e-13)
    with pytest.raises(ValueError):
        G.corona.sky_rayset(G.KerrMetric(1.0, 0.0), G.LampPostModel(h=2.5), G.EvenSampler(), 8)


def test_rows_of_many_samples_at_once_equal_the_rows_sample_by_sample(G):
    """A source without one position brings 28 doubles per sample; the host forms them for all samples at once (positions from the
    model's generator in the order sample_position_velocity would draw them, velocities and Gram-Schmidt tetrads on arrays: 10 µs per
    sample where the loop took 330).  Same draws, same generator state afterwards, rows equal to the last bits of the 4 x 4 products
    -- also for the samples next to the polar axis, where the tetrad's loop (`while sum(p) > tol`) is decided by rounding."""
    K = G.corona
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())

    class OneByOne:            # the same model without sample_positions: sky_rayset asks sample by sample
        point_source = fixed_position = False

        def __init__(self, model):
            self.model = model

        def sample_position_velocity(self, m):
            return self.model.sample_position_velocity(m)

    for m in (G.KerrMetric(1.0, 0.998), G.JohannsenMetric(1.0, 0.6, 1.0, 0.0, 0.0, 0.5)):
        for vf in (G.SourceVelocities.co_rotating, G.SourceVelocities.stationary):
            n = 400 if isinstance(m, G.KerrMetric) else 40
            a, b = G.DiscCorona(vf, 10.0, 5.0, seed=17), G.DiscCorona(vf, 10.0, 5.0, seed=17)
            rows_a = K.sky_rayset(m, a, s, n)[1][1]
            rows_b = K.sky_rayset(m, OneByOne(b), s, n)[1][1]
            np.testing.assert_array_equal(rows_a[:, 0:4], rows_b[:, 0:4])
            np.testing.assert_allclose(rows_a, rows_b, rtol=1e-14, atol=1e-14 * np.abs(rows_b).max())
            assert a.rng.bit_generator.state == b.rng.bit_generator.state
            assert np.min(rows_a[:, 2]) < 0.05                       # samples next to the axis are among them
    # the record route's arrays come from the same rows
    xs, vs, vsrc = K.sample_position_direction_velocity(G.KerrMetric(1.0, 0.9), G.DiscCorona(G.SourceVelocities.co_rotating, 6.0, 4.0, seed=3), s, 300)
    xo, vo, vso = K.sample_position_direction_velocity(G.KerrMetric(1.0, 0.9), OneByOne(G.DiscCorona(G.SourceVelocities.co_rotating, 6.0, 4.0, seed=3)), s, 300)
    np.testing.assert_array_equal(xs, xo)
    np.testing.assert_allclose(vs, vo, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vsrc, vso, rtol=1e-14)
    # a source that lies inside 1.9 inner radii altogether is refused (the loop used to redraw for ever)
    with pytest.raises(ValueError, match="inside 1.9 inner radii"):
        K.sky_rayset(G.JohannsenMetric(1.0, 0.6, 1.0, 0.0, 0.0, 0.5), G.DiscCorona(G.SourceVelocities.co_rotating, 3.0, 1.5, seed=1), s, 10)


# ---------------- gpu ----------------
@pytest.mark.gpu
def test_sky_rays_start_as_the_host_sampler_says(G, ens):
    """Every sampler x generator x domain: the device's (x_init, v_init) and where the rays end against the same rays
    formed by numpy and handed over as arrays."""
    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    for name, m, model, d in _models(G):
        for mk in _samplers(G):
            s_dev, s_host = mk(), mk()                      # two generators with the same seed
            n = 512
            got = K.tracegeodesics(m, model, d, (0.0, 2000.0), n_samples=n, sampler=s_dev, ensemble=ens)
            xs, vs, _ = K.sample_position_direction_velocity(m, model, s_host, n)
            ref = G.tracegeodesics(m, xs, vs, d, (0.0, 2000.0), ensemble=ens)
            np.testing.assert_array_equal(got["x_init"], ref["x_init"])
            np.testing.assert_allclose(got["v_init"], ref["v_init"], rtol=1e-14, atol=2e-14, err_msg=name)
            same = got["status"] == ref["status"]
            assert same.sum() >= n - 1, name
            hit = same & (ref["status"] == G.StatusCodes.IntersectedWithGeometry)
            np.testing.assert_allclose(got["x"][hit], ref["x"][hit], rtol=1e-9, atol=1e-9, err_msg=name)
            np.testing.assert_allclose(got["v"][hit], ref["v"][hit], rtol=1e-8, atol=1e-9, err_msg=name)


@pytest.mark.gpu
def test_corona_rows_equal_oracle_traced_rays_and_their_energy_ratio(G, ens, oracle):
    """The (g, ρ, t, status) rows gr_corona_trace reduces -- here through gr_ray_summary, the call it is built on -- against rays the
    ORACLE traces from the host sampler's (x, v), with energy_ratio (flux-calculations.jl:96-110) formed in numpy from the oracle's
    end points, the source's four-velocity and the Keplerian disc velocity: nothing on the reference side of this comparison has
    been near the device."""
    import ctypes as C

    from gradus_jl_amd import _lib
    from gradus_jl_amd.pointfunctions import GR_PF_REDSHIFT, PointFunction
    from gradus_jl_amd.rendering import abi_pointfunction
    from gradus_jl_amd.tracing import tracing_configuration

    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    a = 0.9
    m, model, d = G.KerrMetric(1.0, a), G.LampPostModel(h=8.0), G.ThinDisc(0.0, 300.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    n = 3000
    rs, keep, x, v_src = K.sky_rayset(m, model, s, n)
    config = tracing_configuration(m, x, np.zeros((1, 4)), d, (0.0, 5000.0), callback=G.domain_upper_hemisphere(), ensemble=ens)
    plunging = K._plunging_table(m, ens)
    pf, kp = abi_pointfunction(PointFunction(None, device_pf=GR_PF_REDSHIFT, extra={"r_isco": m.isco(), "plunge": plunging}))
    pf.has_u_src = 1
    for q in range(4):
        pf.u_src[q] = v_src[q]
    cfg = config.abi_config()
    rows = np.zeros((n, 4))
    _lib.check(_lib.load().gr_ray_summary(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), rows.ctypes.data, None))
    # the oracle on the host sampler's rays
    xs, vs, vsrc = K.sample_position_direction_velocity(m, model, s, n)
    ocfg = oracle.make_config("kerr", (1.0, a), disc=(0.0, 300.0), lambda_max=5000.0, upper_hemisphere=True)
    ref = oracle.trace(ocfg, xs[0], vs)
    st = rows[:, 3].astype(int)
    assert int(np.sum(st != ref["status"])) <= 2
    hit = (st == G.StatusCodes.IntersectedWithGeometry) & (ref["status"] == G.StatusCodes.IntersectedWithGeometry)
    rho = ref["x"][:, 1] * np.abs(np.sin(ref["x"][:, 2]))
    out = hit & (rho > 1.05 * m.isco())            # Keplerian disc: circular_fourvelocity is closed-form host code
    assert out.sum() > 1000
    np.testing.assert_allclose(rows[hit, 1], rho[hit], rtol=1e-6)
    np.testing.assert_allclose(rows[hit, 2], ref["x"][hit, 0], rtol=1e-6)
    v_disc = K.circular_fourvelocity(m, rho[out])
    g_ref = K.energy_ratio(m, ref[out], vsrc[0], v_disc)
    np.testing.assert_allclose(rows[out, 0], g_ref, rtol=1e-6)


@pytest.mark.gpu
def test_device_radial_profile_equals_the_record_route(G, ens, monkeypatch):
    """emissivity_profile with the per-ray half on the device against tracecorona + build_radial_profile on the same samples."""
    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    for name, m, model, d in _models(G):
        for mk in (lambda: G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator()),
                   lambda: G.WeierstrassSampler(200.0, G.LowerHemisphere(), G.RandomGenerator(seed=5))):
            kw = dict(n_samples=30_000, N=40, ensemble=ens)
            monkeypatch.setenv("GRADUS_MI355X_DEVICE_CORONA", "1")
            dev = G.emissivity_profile(m, d, model, sampler=mk(), **kw)
            monkeypatch.setenv("GRADUS_MI355X_DEVICE_CORONA", "0")
            host = G.emissivity_profile(m, d, model, sampler=mk(), **kw)
            np.testing.assert_allclose(dev.radii, host.radii, rtol=1e-10, err_msg=name)
            # The last edge of grid(extrema(radii)..., N) is the largest radius itself up to rounding, so whether that one ray
            # counts for the last bin or the one before is decided by the last bit of ρ_max in the reference too: the two
            # outermost bins are compared together, the others one by one.
            inner = slice(0, -2)
            ok = np.isfinite(host.ε[inner])
            np.testing.assert_array_equal(np.isfinite(dev.ε[inner]), ok, err_msg=name)
            assert ok.sum() >= 20, name
            np.testing.assert_allclose(dev.ε[inner][ok], host.ε[inner][ok], rtol=1e-7, err_msg=name)
            okt = np.isfinite(host.t[inner])
            np.testing.assert_array_equal(np.isfinite(dev.t[inner]), okt, err_msg=name)
            np.testing.assert_allclose(dev.t[inner][okt], host.t[inner][okt], rtol=1e-9, err_msg=name)
            assert np.isfinite(dev.ε[-2:]).any() and np.isfinite(host.ε[-2:]).any()


@pytest.mark.gpu
def test_disc_corona_on_the_device_equals_the_record_route(G, ens, monkeypatch):
    """A source without one position (DiscCorona) through gr_corona_trace / gr_corona_bin -- 28 doubles per sample in, the per-sample
    energy ratio formed on the device -- against tracecorona + build_radial_profile on records of the SAME draws (two generators
    with one seed), and the end-point records of the sky rays against the host sampler's rays."""
    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    m, d = G.KerrMetric(1.0, 0.9), G.ThinDisc(0.0, 200.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    mk = lambda: G.DiscCorona(G.SourceVelocities.co_rotating, 6.0, 4.0, seed=17)
    n = 4000
    # the rays themselves: device-formed against host-formed
    got = K.tracegeodesics(m, mk(), d, (0.0, 3000.0), n_samples=n, sampler=s, ensemble=ens)
    xs, vs, _ = K.sample_position_direction_velocity(m, mk(), s, n)
    ref = G.tracegeodesics(m, xs, vs, d, (0.0, 3000.0), ensemble=ens)
    np.testing.assert_array_equal(got["x_init"], ref["x_init"])
    np.testing.assert_allclose(got["v_init"], ref["v_init"], rtol=1e-13, atol=1e-13)
    assert (got["status"] == ref["status"]).sum() >= n - 2
    # the profile
    dev = K.emissivity_profile(m, d, mk(), n_samples=n, sampler=s, N=40, ensemble=ens)
    monkeypatch.setenv("GRADUS_MI355X_DEVICE_CORONA", "0")
    rec = K.emissivity_profile(m, d, mk(), n_samples=n, sampler=s, N=40, ensemble=ens)
    monkeypatch.delenv("GRADUS_MI355X_DEVICE_CORONA")
    np.testing.assert_allclose(dev.radii, rec.radii, rtol=1e-12)
    # (The grid's last edge IS the largest hit radius, rounded by the grid's own arithmetic: the one ray that defines it lands in
    # the last bin or in the one before by a bit of that rounding -- the two routes' rays differ in their last bits -- so the last
    # two bins are compared as one.)
    ok = np.isfinite(rec.ε)
    np.testing.assert_array_equal(np.isfinite(dev.ε)[:-1], ok[:-1])
    ok[-2:] = False
    assert ok.sum() > 25
    np.testing.assert_allclose(dev.ε[ok], rec.ε[ok], rtol=1e-9)
    np.testing.assert_allclose(dev.t[ok], rec.t[ok], rtol=1e-10)


@pytest.mark.gpu
def test_rays_dealt_by_cost_are_a_permutation_of_the_samples(G, ens):
    """gr_corona_trace deals its sky rays to the waves by what they will cost (counting sort over (class, chunk), k_sky_velocities_dealt):
    every sample exactly once, whatever the order -- the integer bins of the dealt launch are the BITS of the launch in sample order
    (knob sky_deal = 0), for a source at one position and for one with a position per sample, at a sample count that is not a
    multiple of the chunk."""
    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    m, d = G.KerrMetric(1.0, 0.9), G.ThinDisc(0.0, 200.0)
    for mk in (lambda: G.LampPostModel(h=7.0), lambda: G.DiscCorona(G.SourceVelocities.co_rotating, 6.0, 4.0, seed=3)):
        prof = {}
        for deal in (1, 0):
            ens.set("sky_deal", deal)
            prof[deal] = K.device_radial_profile(m, d, mk(), sampler=s, n_samples=50_001, N=60, ensemble=ens)
        ens.set("sky_deal", 1)
        np.testing.assert_array_equal(prof[1].radii, prof[0].radii)
        np.testing.assert_array_equal(prof[1].ε, prof[0].ε)
        np.testing.assert_array_equal(prof[1].t, prof[0].t)
        assert np.isfinite(prof[1].ε).sum() > 40
    # ... and every context of an ensemble deals its own share (three shares of 20 000 samples each, all of them dealt): the same bits
    ens3 = G.EnsembleMI355X(devices=[0, 0, 0])
    one = K.device_radial_profile(m, d, G.LampPostModel(h=7.0), sampler=s, n_samples=60_000, N=60, ensemble=ens)
    three = K.device_radial_profile(m, d, G.LampPostModel(h=7.0), sampler=s, n_samples=60_000, N=60, ensemble=ens3)
    np.testing.assert_array_equal(three.ε, one.ε)
    np.testing.assert_array_equal(three.t, one.t)


@pytest.mark.gpu
def test_corona_bins_are_the_bucket_rule_and_sum_to_the_hits(G, ens):
    """gr_corona_trace / gr_corona_bin through the C ABI: counts add up to the hits, re-binning the same trace with other
    edges needs no new trace, edge cases of bucket(Simple()) (below the first edge / above the last)."""
    from gradus_jl_amd import _lib
    from gradus_jl_amd.rendering import abi_pointfunction
    from gradus_jl_amd.pointfunctions import GR_PF_REDSHIFT, PointFunction
    from gradus_jl_amd.tracing import tracing_configuration

    K = G.corona
    m = G.KerrMetric(1.0, 0.998)
    model, d = G.LampPostModel(h=6.0), G.ThinDisc(0.0, 300.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    n = 20_000
    rs, keep, x, v = K.sky_rayset(m, model, s, n)
    config = tracing_configuration(m, x, np.zeros((1, 4)), d, (0.0, 5000.0), callback=G.domain_upper_hemisphere(), ensemble=ens)
    cfg = config.abi_config()
    L = _lib.load()
    # a context that has not traced a corona has nothing to bin
    fresh = G.EnsembleMI355X(device=0)
    edges = np.linspace(1.0, 300.0, 16)
    out = np.zeros((3, edges.size))
    assert L.gr_corona_bin(fresh.ctx.handle, edges.ctypes.data, edges.size, out.ctypes.data) != 0
    assert "gr_corona_trace" in _lib.load().gr_last_error().decode()
    pf, kp = abi_pointfunction(PointFunction(None, device_pf=GR_PF_REDSHIFT, extra={"r_isco": m.isco(), "plunge": None}))
    pf.has_u_src = 1
    for q in range(4):
        pf.u_src[q] = v[q]
    lim, hits, st = np.zeros(2), C.c_int64(0), _lib.gr_stats()
    _lib.check(L.gr_corona_trace(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), lim.ctypes.data, C.byref(hits), C.byref(st)))
    assert 0.3 * n < hits.value < 0.7 * n and 1.0 < lim[0] < 1.5 and 250.0 < lim[1] <= 300.0
    # the same rays as records
    gps = K.tracegeodesics(m, model, d, (0.0, 5000.0), n_samples=n, sampler=s, ensemble=ens, callback=G.domain_upper_hemisphere())
    hit = gps["status"] == G.StatusCodes.IntersectedWithGeometry
    rho = gps["x"][hit, 1] * np.abs(np.sin(gps["x"][hit, 2]))
    assert hits.value == hit.sum()
    assert lim[0] == rho.min() and lim[1] == rho.max()
    for edges in (np.linspace(lim[0], lim[1], 32), np.array([5.0, 10.0, 20.0]), np.array([2.0])):
        edges = np.ascontiguousarray(edges)
        out = np.zeros((3, edges.size))
        _lib.check(L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, out.ctypes.data))
        idx = K._bucket_index(rho, edges)
        np.testing.assert_array_equal(out[0], np.bincount(idx, minlength=edges.size))
        np.testing.assert_allclose(out[2], np.bincount(idx, weights=gps["x"][hit, 0], minlength=edges.size), rtol=1e-12)
        assert out[0].sum() == hits.value and np.all(out[1][out[0] > 0] > 0)
        again = np.zeros_like(out)
        _lib.check(L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, again.ctypes.data))
        np.testing.assert_array_equal(again, out)                     # integer accumulation: the same bits every time
    # The rows of a trace live in a buffer of their own: OTHER work on the context between gr_corona_trace and gr_corona_bin -- an
    # image, a ray set (both stage tables, use the stats block, the sky buffer and the result buffers) -- leaves them as they are ...
    edges = np.ascontiguousarray(np.linspace(lim[0], lim[1], 32))
    before = np.zeros((3, edges.size))
    _lib.check(L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, before.ctypes.data))
    xo = np.array([0.0, 1000.0, 1.2, 0.0])
    G.rendergeodesics(m, xo, G.ThinDisc(m.isco(), 40.0), 2000.0, image_width=64, image_height=64, alpha_lims=(-30, 30), beta_lims=(-20, 20),
                      pf=G.ConstPointFunctions.redshift(m, xo) @ G.ConstPointFunctions.filter_intersected(), ensemble=ens)
    K.tracegeodesics(m, model, d, (0.0, 5000.0), n_samples=500, sampler=s, ensemble=ens, callback=G.domain_upper_hemisphere())
    after = np.zeros_like(before)
    _lib.check(L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, after.ctypes.data))
    np.testing.assert_array_equal(after, before)
    # ... and a gr_corona_trace that is refused leaves NO rows behind: the next gr_corona_bin fails instead of binning the previous trace's
    bad_cfg = type(cfg).from_buffer_copy(cfg)
    bad_cfg.abstol = -1.0
    assert L.gr_corona_trace(ens.ctx.handle, C.byref(bad_cfg), C.byref(rs), C.byref(pf), lim.ctypes.data, C.byref(hits), C.byref(st)) != 0
    assert L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, after.ctypes.data) != 0
    assert "gr_corona_trace" in _lib.load().gr_last_error().decode()
    _lib.check(L.gr_corona_trace(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), lim.ctypes.data, C.byref(hits), C.byref(st)))
    # more bins than the LDS histogram holds: global atomics, the same sums
    edges = np.ascontiguousarray(np.geomspace(lim[0], lim[1], 3000))
    out = np.zeros((3, edges.size))
    _lib.check(L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, out.ctypes.data))
    idx = K._bucket_index(rho, edges)
    np.testing.assert_array_equal(out[0], np.bincount(idx, minlength=edges.size))
    np.testing.assert_allclose(out[2], np.bincount(idx, weights=gps["x"][hit, 0], minlength=edges.size), rtol=1e-13)
    # descending edges are refused
    bad = np.array([3.0, 2.0])
    assert L.gr_corona_bin(ens.ctx.handle, bad.ctypes.data, 2, out.ctypes.data) != 0
    # A sky source shards over contexts like any ray set (ABI 8: gr_rayset.sky_first / sky_total -- the sample numbers of a share go
    # on counting where the previous share stopped): the same rows from three contexts as from one, ...
    third = G.EnsembleMI355X(0)
    three = [ens.ctx, fresh.ctx, third.ctx]
    arr, sts = _lib.ctx_array(three)
    rows1, rows3 = np.zeros((n, 4)), np.zeros((n, 4))
    _lib.check(L.gr_ray_summary(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), rows1.ctypes.data, None))
    _lib.check(L.gr_ray_summary_multi(arr, 3, C.byref(cfg), C.byref(rs), C.byref(pf), rows3.ctypes.data, sts))
    np.testing.assert_array_equal(rows3, rows1)
    assert sum(x.rays for x in sts) == n and all(x.rays > 0 for x in sts)
    # ... and gr_corona_trace_multi / gr_corona_bin_multi give the limits, the hit count and -- integer accumulators on ONE grid
    # add up exactly -- the BITS of the one-context calls
    _lib.check(L.gr_corona_trace(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), lim.ctypes.data, C.byref(hits), C.byref(st)))
    one = np.zeros((3, 32))
    edges = np.ascontiguousarray(np.linspace(lim[0], lim[1], 32))
    _lib.check(L.gr_corona_bin(ens.ctx.handle, edges.ctypes.data, edges.size, one.ctypes.data))
    lim3, hits3 = np.zeros(2), C.c_int64(0)
    _lib.check(L.gr_corona_trace_multi(arr, 3, C.byref(cfg), C.byref(rs), C.byref(pf), lim3.ctypes.data, C.byref(hits3), sts))
    assert hits3.value == hits.value and lim3[0] == lim[0] and lim3[1] == lim[1]
    multi = np.zeros((3, 32))
    _lib.check(L.gr_corona_bin_multi(arr, 3, edges.ctypes.data, edges.size, multi.ctypes.data))
    np.testing.assert_array_equal(multi, one)
    # contexts that do not hold the shares of one trace are refused
    rs_other = type(rs).from_buffer_copy(rs)
    rs_other.n = n // 3
    _lib.check(L.gr_corona_trace(third.ctx.handle, C.byref(cfg), C.byref(rs_other), C.byref(pf), lim.ctypes.data, C.byref(hits), C.byref(st)))
    assert L.gr_corona_bin_multi(arr, 3, edges.ctypes.data, edges.size, multi.ctypes.data) != 0
    assert "ONE gr_corona_trace_multi" in _lib.load().gr_last_error().decode()
    # the Python route: an ensemble over several contexts
    ens3 = G.EnsembleMI355X(devices=[0, 0, 0])
    pa = K.device_radial_profile(m, d, model, sampler=s, n_samples=n, N=40, ensemble=ens)
    pb = K.device_radial_profile(m, d, model, sampler=s, n_samples=n, N=40, ensemble=ens3)
    np.testing.assert_array_equal(pb.ε, pa.ε)
    np.testing.assert_array_equal(pb.t, pa.t)


@pytest.mark.gpu
def test_corona_of_a_tabulated_metric_equals_the_fused_one(G, ens):
    """A user-defined metric (GR_METRIC_TABULATED) under a corona: sky rays, energy ratio and bins through the table."""
    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    kerr = G.KerrMetric(1.0, 0.9)
    tab = G.TabulatedMetric(kerr)
    d = G.ThinDisc(0.0, 300.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    for model in (G.LampPostModel(h=8.0), G.RingCorona(G.SourceVelocities.co_rotating, 6.0, 4.0)):
        a = K.device_radial_profile(kerr, d, model, sampler=s, n_samples=20_000, N=30, ensemble=ens)
        b = K.device_radial_profile(tab, d, model, sampler=s, n_samples=20_000, N=30, ensemble=ens)
        np.testing.assert_allclose(b.radii, a.radii, rtol=1e-7)
        inner = slice(0, -2)
        ok = np.isfinite(a.ε[inner])
        np.testing.assert_array_equal(np.isfinite(b.ε[inner]), ok)
        # outside the ISCO the disc is Keplerian (first derivatives of the table, 1e-8); inside, its velocity is the traced plunge,
        # which starts from a circular orbit at the ISCO where E² - V_eff cancels to ~0: the table's 1e-8 shows as ~1e-5 there
        out = (a.radii[inner] > kerr.isco())[ok]
        assert out.sum() >= 15 and (~out).sum() >= 2
        np.testing.assert_allclose(b.ε[inner][ok][out], a.ε[inner][ok][out], rtol=1e-6)
        np.testing.assert_allclose(b.ε[inner][ok][~out], a.ε[inner][ok][~out], rtol=2e-4)
        np.testing.assert_allclose(b.t[inner][ok], a.t[inner][ok], rtol=1e-7)


@pytest.mark.gpu
def test_a_million_samples_stay_on_the_device(G, ens):
    """the per-ray half of emissivity_profile at 10⁶ samples.  Timing is reported, the
    assertion is loose: the device work of the call is tens of milliseconds, not the seconds of the record route."""
    K = G.corona
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    model, d = G.LampPostModel(h=10.0), G.ThinDisc(0.0, 500.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    K.device_radial_profile(m, d, model, sampler=s, n_samples=10_000, N=100, ensemble=ens)        # warm: plunging table, contexts
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        prof, st = K.device_radial_profile(m, d, model, sampler=s, n_samples=1_000_000, N=100, ensemble=ens, stats=True)
        wall = (time.perf_counter() - t0) * 1e3
        best = (st.kernel_ms, st.call_ms, wall) if best is None or st.call_ms < best[1] else best
    print(f"corona 1e6 samples: trace kernel {best[0]:.2f} ms, device call {best[1]:.2f} ms, wall {best[2]:.1f} ms")
    assert np.isfinite(prof.ε).sum() >= 90 and best[1] < 100.0
    # and it is the profile a 50x smaller run gives, to Monte-Carlo noise (ε counts photons: it scales with n_samples, as in the reference)
    small = K.device_radial_profile(m, d, model, sampler=s, n_samples=20_000, N=100, ensemble=ens)
    r = np.geomspace(2.0, 200.0, 12)
    np.testing.assert_allclose(small.emissivity_at(r) * 50.0, prof.emissivity_at(r), rtol=0.25)


@pytest.mark.gpu
def test_corona_through_the_fp32_kernels(G, ens):
    """`precision 32` (the kernels of the C5 tolerance sweep) under a corona: the sky rays arrive as fp64 arrays, the energy ratio
    against a moving source is evaluated in single precision -- the profile follows the fp64 one to the tolerance asked."""
    K = G.corona
    m = G.KerrMetric(1.0, 0.9)
    d = G.ThinDisc(0.0, 300.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    for model in (G.LampPostModel(h=8.0), G.BeamedPointSource(10.0, 0.4)):
        ens.set("kernel", 2).set("precision", 64)
        a = K.device_radial_profile(m, d, model, sampler=s, n_samples=40_000, N=30, ensemble=ens)
        ens.set("precision", 32)
        try:
            b = K.device_radial_profile(m, d, model, sampler=s, n_samples=40_000, N=30, ensemble=ens, abstol=1e-5, reltol=1e-5)
        finally:
            ens.set("precision", 64)
        # (ρ_min is the innermost hit, a ray that skims the horizon: the edges built from it move by its fp32 noise, and a few
        # photons change bins with them)
        np.testing.assert_allclose(b.radii, a.radii, rtol=3e-2)
        # Compared beyond r = 16: at tolerance 1e-5 the steps near the hole are longer than the disc's slab (half-thickness gtol r)
        # is thick, and the callback's eight samples per step miss crossings there -- in fp64 at 1e-5 exactly as in fp32, and in the
        # reference's ContinuousCallback by the same mechanism (a third of the photons inside r = 5); not a property of the kernels
        r = np.geomspace(16.0, 0.5 * a.radii[-1], 10)
        np.testing.assert_allclose(b.emissivity_at(r), a.emissivity_at(r), rtol=3e-2)
        np.testing.assert_allclose(b.coordtime_at(r), a.coordtime_at(r), rtol=1e-3)
        ens.set("precision", 64)
        c = K.device_radial_profile(m, d, model, sampler=s, n_samples=40_000, N=30, ensemble=ens, abstol=1e-5, reltol=1e-5)
        ri = np.array([3.0, 4.0, 6.0])
        np.testing.assert_allclose(b.emissivity_at(ri), c.emissivity_at(ri), rtol=0.4)       # fp32 and fp64 at the SAME tolerance lose the same share there (which photons: a matter of step placement)
