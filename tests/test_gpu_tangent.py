"""`gr_ray_tangent` on the MI355X: the TANGENT flavour of the kernels (real = value + ∂/∂α + ∂/∂β through the
integrator; SURVEY §8 f-4, jacobian_∂αβ_∂gr, src/tracing/precision-solvers.jl:401-451) against (1) its own host build,
(2) central differences of the fp64 device path, (3) the values the reference records for its Cunningham transfer
functions, with the reference's Newton root finder."""
import math

import numpy as np
import pytest

import harness as Hh

pytestmark = pytest.mark.gpu

GOLD = {(3, 4.0): 0.14048899037409682, (35, 4.0): 0.10846177995555085, (74, 4.0): 0.05550300700779827,
        (85, 4.0): 0.03602870590038378, (30, 4.0): 0.11958152396826184, (30, 7.0): 0.12205125501900763,
        (30, 10.0): 0.1265019201038228, (30, 15.0): 0.12875961522283233, (30, 300.0): 0.13378948600255888,
        (30, 800.0): 0.13470290875241375, (30, 1000.0): 0.13319637850028626}
# the three recorded statistics this build does not reproduce (f-4): kept in the table as STRICT xfails so that the gap is
# visible in every GPU test record; scripts/tf_truncation_scan.py has the lead on their origin
UNMET = {(3, 4.0), (30, 4.0), (35, 4.0)}


# Both shapes of the tangent kernels (kernels_tu.hip): "lane" = one lane carries a ray with both directions of the Jacobian
# (the throughput shape), "pairs" = a pair of neighbouring lanes per ray, one direction each (the latency shape; the Dual
# norm's sums cross the pair).  Every test of this file runs on both; by default the library picks by launch size.
@pytest.fixture(params=[0, 1], ids=["lane", "pairs"])
def ens(G, request):
    return G.EnsembleMI355X(0, tangent_pairs=request.param)


def test_the_two_tangent_shapes_agree(G):
    """Same rays through both shapes and through the library's own choice: values, tangents and status agree to rounding
    (the value part is the same arithmetic; the tangents are the same formulas on one or two members)."""
    x = np.array([0.0, 1000.0, math.radians(30), 0.0])
    m = G.KerrMetric(1.0, 0.998)
    rng = np.random.default_rng(3)
    al, be = rng.uniform(-12, 12, 1000), rng.uniform(-12, 12, 1000)
    out = {}
    for shape in (0, 1, 2):
        out[shape] = _tracer(G, G.EnsembleMI355X(0, tangent_pairs=shape), m, x, 4000.0).tangent(al, be)
    hit = out[0][:, 7] == 2
    assert hit.sum() > 600 and np.array_equal(out[0][:, 7], out[1][:, 7])
    np.testing.assert_allclose(out[1][hit, 0:2], out[0][hit, 0:2], rtol=1e-11)
    for c in range(2, 6):
        scale = np.abs(out[0][hit, c]).max()
        assert np.abs(out[1][hit, c] - out[0][hit, c]).max() < 1e-9 * scale, c
    assert out[2].tobytes() == out[1].tobytes()          # 1000 rays: the library takes the pairs


def _tracer(G, ens, m, x, max_time, **kw):
    from gradus_jl_amd.transfer_functions import device_tracer

    chart = G.chart_for_metric(m, 2 * x[1], closest_approach=1.005)
    return device_tracer(m, x, max_time, chart, G.ConstPointFunctions.redshift(m, x), ens, **kw)


def test_device_tangents_equal_the_host_build(G, ens):
    """Same source, two compilers: values to 1e-10, tangents to 1e-7 of their scale (contraction differs)."""
    x = np.array([0.0, 1000.0, math.radians(30), 0.0])
    m = G.KerrMetric(1.0, 0.998)
    rng = np.random.default_rng(5)
    al, be = rng.uniform(-12, 12, 200), rng.uniform(-12, 12, 200)
    tr = _tracer(G, ens, m, x, 4000.0)
    dev = tr.tangent(al, be)
    cfg = G.tracing_configuration(m, x, np.zeros((1, 4)), G.DatumPlane(0.0), 4000.0,
                                  chart=G.chart_for_metric(m, 2 * x[1], closest_approach=1.005))
    host = Hh.ray_tangent(G, cfg, G.ConstPointFunctions.redshift(m, x), al, be)      # both sides: tangents in the norm (default)
    assert np.array_equal(dev[:, 7], host[:, 7])
    hit = dev[:, 7] == 2
    assert hit.sum() > 120
    np.testing.assert_allclose(dev[hit, 0:2], host[hit, 0:2], rtol=1e-9)
    for c in range(2, 6):
        scale = np.abs(host[hit, c]).max()
        assert np.abs(dev[hit, c] - host[hit, c]).max() < 1e-6 * scale, c
    # and the value part is the plain kernel's answer: to rounding when the controller looks at values only (same steps),
    # to the tolerance level with the tangents in the norm (the default: shorter steps where the tangents are stiff)
    pts, g = tr(al, be)
    np.testing.assert_allclose(dev[hit, 0], g[hit], rtol=1e-5)
    np.testing.assert_allclose(dev[hit, 1], pts["x"][hit, 1], rtol=1e-5)
    ens.set("tangent_norm", 0)
    try:
        same = _tracer(G, ens, m, x, 4000.0).tangent(al, be)
    finally:
        ens.set("tangent_norm", 1)
    np.testing.assert_allclose(same[hit, 0], g[hit], rtol=1e-8)
    np.testing.assert_allclose(same[hit, 1], pts["x"][hit, 1], rtol=1e-8)


@pytest.mark.parametrize("name", ["kerr", "johannsen", "kerr-newman", "johannsen-psaltis", "bumblebee", "dilaton-axion"])
def test_device_tangents_equal_central_differences(G, ens, name):
    """Per metric family (the hand-fused right-hand sides evaluated on value + tangent scalars: Kerr, Johannsen, the Kerr core with
    an r-dependent mass / index, the closed-determinant forms): the tangents against central differences of the fp64 device path
    at tolerance 1e-12."""
    m = {"kerr": G.KerrMetric(1.0, 0.9), "johannsen": G.JohannsenMetric(1.0, 0.7, 1.0, 0.0, 0.0, 0.5),
         "kerr-newman": G.KerrNewmanMetric(1.0, 0.6, 0.5), "johannsen-psaltis": G.JohannsenPsaltisMetric(1.0, 0.6, 1.0),
         "bumblebee": G.BumblebeeMetric(1.0, 0.2, 0.3), "dilaton-axion": G.DilatonAxion(1.0, 0.5, 0.2, 0.8)}[name]
    # (Kerr-dark-matter and Kerr-refractive are not in the list: central differences through the kinks of the enclosed mass and
    # through the 1e-4-wide index step are no reference at this bound -- 6.1e-5 of the scale for Kerr-refractive with its fused
    # right-hand side, 5.1e-5 with the dual-number one, against the 5e-5 asked here; their fused forms are pinned at the level
    # of the right-hand side, tests/test_kernel_logic_host.py)
    x = np.array([0.0, 1000.0, math.radians(55), 0.0])
    rng = np.random.default_rng(11)
    al, be = rng.uniform(-10, 10, 64), rng.uniform(3, 12, 64) * rng.choice([-1, 1], 64)
    tr = _tracer(G, ens, m, x, 4000.0, abstol=1e-12, reltol=1e-12)
    t = tr.tangent(al, be)
    h = 1e-5
    (pa, ga), (ma, gma) = tr(al + h, be), tr(al - h, be)
    (pb, gb), (mb, gmb) = tr(al, be + h), tr(al, be - h)
    ok = (t[:, 7] == 2) & (pa["status"] == 2) & (ma["status"] == 2) & (pb["status"] == 2) & (mb["status"] == 2)
    ok &= np.isfinite(ga) & np.isfinite(gma) & np.isfinite(gb) & np.isfinite(gmb) & np.isfinite(t[:, 0])
    ok &= t[:, 1] > 1.3 * m.isco()                      # the redshift inside the ISCO needs the plunging table: not this test
    assert ok.sum() > 30
    with np.errstate(all="ignore"):          # (rays outside `ok` may hold infinities)
        fd = np.stack([(ga - gma), (gb - gmb), pa["x"][:, 1] - ma["x"][:, 1], pb["x"][:, 1] - mb["x"][:, 1]], axis=1) / (2 * h)
    for c in range(4):
        scale = np.abs(fd[ok, c]).max()
        err = np.abs(t[ok, 2 + c] - fd[ok, c]).max()
        assert err < 5e-5 * scale, (name, c, err, scale)


def _gold_params():
    out = []
    for (angle, r) in sorted(GOLD):
        marks = [pytest.mark.xfail(strict=True, reason="recorded statistic not reproduced with the 114 samples the reference's "
                                   "current source stores (f-4)")] if (angle, r) in UNMET else []
        out.append(pytest.param(angle, r, marks=marks, id=f"{angle}deg-re{r:g}"))
    return out


@pytest.mark.parametrize("angle,r", _gold_params())
def test_reference_transfer_function_values_with_dual_numbers(G, ens, angle, r):
    """ALL ELEVEN recorded statistics (test/smoke-tests/cunningham-transfer-functions.jl:25-39; atol 1e-3, rtol 1e-2 for the
    three large radii) with the reference's root finder and dual-number Jacobians on the device.  Eight are met -- six of
    them two to four digits inside the reference's tolerance --, three are strict xfails."""
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 100_000.0, math.radians(angle), 0.0])
    c = G.cunningham_transfer_function(m, x, G.ThinDisc(0.0, float("inf")), r, N=80, ensemble=ens, root_finder="reference",
                                       chart=G.chart_for_metric(m, 2 * x[1], closest_approach=1.005))
    assert c.f.size == 114 and np.all(np.isfinite(c.f))
    meas = float(np.sum(c.f * c.g_star) / c.f.size)
    print(f"  ({angle}°, {r}): {meas - GOLD[(angle, r)]:+.2e}")
    if (angle, r) in UNMET:
        assert meas == pytest.approx(GOLD[(angle, r)], abs=1e-3)          # the reference's own bound: expected to fail
        return
    # the statistic moves by 1e-4 ... 3e-4 per 1e-10 of relative noise in g (its extremal samples), so two builds
    # of the same integrator (host g++ / device hipcc contraction) differ at the 1e-4 level: 3e-4 where the host build is
    # within 1e-4, the reference's 1e-3 for (30°, 15) [+4.1e-4] and (85°, 4) [-5.5e-4]
    tol = 1e-3 if (angle, r) in ((30, 15.0), (85, 4.0)) else 3e-4
    assert meas == pytest.approx(GOLD[(angle, r)], abs=tol)


def test_device_tangents_equal_the_tangent_oracle_ray_by_ray(G, ens, oracle):
    """gr_ray_tangent pinned PER RAY against the oracle's dual-number mode (oracle/tangent_oracle.cpp: the unchanged oracle
    source on a value + ∂/∂α + ∂/∂β scalar -- a second integrator: first-order form, library sin/cos, true divisions, its
    own event root find and its own event-time term).  With the tangents in the error norm (default; DiffEqBase's norm on
    Dual state) values agree to 1e-5 (5e-6 measured) and Jacobian entries to 1e-5 of their scale (1e-6 measured) on 300 rays of the transfer-function
    geometry; with values-only control (knob 0) ordinary rays agree to 1e-4 and the Jacobian is unprotected near the
    polar axis."""
    a = 0.998
    m = G.KerrMetric(1.0, a)
    x = np.array([0.0, 100_000.0, math.radians(30), 0.0])
    rng = np.random.default_rng(2026)
    rr, th = rng.uniform(2.5, 14.0, 300), rng.uniform(0.0, 2 * math.pi, 300)
    th[:4] = [math.pi / 2, math.pi / 2 + 1e-3, 3 * math.pi / 2, 0.0]         # through / next to the polar axis, and the α axis
    al, be = rr * np.cos(th), rr * np.sin(th)
    cfg = oracle.make_config("kerr", (1.0, a), disc={"datum": 0.0}, lambda_max=2 * x[1], closest_approach=1.005,
                             outer_radius=2 * x[1])

    def rel(dev, orc):
        out = np.zeros(dev.shape[0])
        for lo in (2, 4):
            scale = np.maximum(np.abs(orc[:, lo]), np.abs(orc[:, lo + 1]))
            out = np.maximum(out, np.max(np.abs(dev[:, lo:lo + 2] - orc[:, lo:lo + 2]), axis=1) / scale)
        return out

    res = {}
    for norm in (1, 0):
        ens.set("tangent_norm", norm)
        try:
            dev = _tracer(G, ens, m, x, 2 * x[1]).tangent(al, be)
        finally:
            ens.set("tangent_norm", 1)
        orc = oracle.ray_tangent(cfg, x, al, be, r_isco=m.isco(), max_time=2 * x[1], norm_with_tangents=bool(norm))
        assert np.array_equal(dev[:, 7], orc[:, 7])
        hit = dev[:, 7] == 2
        assert hit.sum() > 250
        np.testing.assert_allclose(dev[hit, 0:2], orc[hit, 0:2], rtol=1e-5)
        res[norm] = rel(dev[hit], orc[hit])
    print(f"  per-ray Jacobian deviation from the tangent oracle: norm on max {res[1].max():.2e} median {np.median(res[1]):.2e}; "
          f"values-only max {res[0].max():.2e} median {np.median(res[0]):.2e}")
    assert res[1].max() < 1e-5 and np.median(res[1]) < 2e-6
    assert np.median(res[0]) < 1e-4


def test_tangent_entry_point_edges(G, ens):
    """Empty ray set; no geometry -> error (tangents are taken where the ray meets it); the context's fp32 switch does not
    apply (tangents are always fp64)."""
    from gradus_jl_amd.transfer_functions import device_tracer

    x = np.array([0.0, 1000.0, math.radians(30), 0.0])
    m = G.KerrMetric(1.0, 0.998)
    tr = _tracer(G, ens, m, x, 4000.0)
    assert tr.tangent(np.zeros(0), np.zeros(0)).shape == (0, 8)
    a, b = np.array([5.0, -3.0]), np.array([4.0, 6.0])
    ref = tr.tangent(a, b)
    ens.set("precision", 32)
    try:
        np.testing.assert_array_equal(_tracer(G, ens, m, x, 4000.0).tangent(a, b), ref)
    finally:
        ens.set("precision", 64)
    # per-ray datum planes (thick-disc transfer functions): heights shift the surface the tangents are taken on
    up = tr.tangent(a, b, heights=np.array([0.5, 0.5]))
    assert np.all(up[:, 7] == 2) and np.all(np.abs(up[:, 1] - ref[:, 1]) > 1e-3)


@pytest.mark.parametrize("angle,re,rel", [(3, 4.0, 2e-4), (30, 4.0, 5e-4), (85, 4.0, 6e-3)])
def test_device_transfer_functions_satisfy_the_normalisation_identity(G, ens, angle, re, rel):
    """∮ (f/g) 2 dφ (g✶ = sin²φ, both branches) = (1/π rₑ) dA/drₑ with A(rₑ) the area enclosed by the image of the ring
    ρ = rₑ -- the transfer function's own normalisation against a number the root finder alone provides; on the device, at
    the points where the reference's recorded mean(f g✶) is not met (tests/test_transfer_functions_tangent_host.py)."""
    from gradus_jl_amd import transfer_functions as TF

    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 100_000.0, math.radians(angle), 0.0])
    chart = G.chart_for_metric(m, 2 * x[1], closest_approach=1.005)
    tr = TF.device_tracer(m, x, 2 * x[1], chart, G.ConstPointFunctions.redshift(m, x), ens)
    th = np.linspace(0.0, 2 * math.pi, 1441)[:-1]

    def area(r_e):
        r = TF.find_offsets_for_radius_newton_ad(tr, np.full(th.size, r_e), th, r_min=m.inner_radius())[0]
        return 0.5 * np.sum(r * r) * (th[1] - th[0])

    dA = (area(re + 1e-3) - area(re - 1e-3)) / 2e-3
    c = G.cunningham_transfer_function(m, x, G.ThinDisc(0.0, float("inf")), re, N=80, chart=chart, ensemble=ens)
    gs = np.clip(c.g_star, 0.0, 1.0)
    y = c.f / (c.gmin + gs * (c.gmax - c.gmin))
    for k in np.flatnonzero(c.f == 0.0):
        y[k] = 0.5 * (y[k - 1] + y[(k + 1) % y.size])
    phi = np.arcsin(np.sqrt(gs))
    total = np.sum(0.5 * (y + np.roll(y, -1)) * 2.0 * np.abs(np.roll(phi, -1) - phi))
    assert total == pytest.approx(dA / (math.pi * re), rel=rel)


def test_thick_disc_transfer_function_satisfies_the_normalisation_identity(G, ens):
    """The second recorded case of test/transfer-functions/test-thick-disc.jl (a = 0.2, 20°, ShakuraSunyaev at Ṁ = 0.2,
    rₑ = 5.47): Σf on the device is 21.4028 at every tolerance, 0.83 % below the recorded 21.5814 -- yet the transfer
    function is correctly normalised: ∮ (f/g) 2 dφ equals (1/π rₑ) dA/drₑ, with A(rₑ) the area enclosed by the image of the
    ring ρ = rₑ ON THE DISC'S SURFACE (offsets against datumplane(d, rₑ)), to 2e-4 with 114 samples and 2e-5 with 434."""
    from gradus_jl_amd import transfer_functions as TF

    m = G.KerrMetric(1.0, 0.2)
    x = np.array([0.0, 10_000.0, math.radians(20), 0.0])
    d = G.ShakuraSunyaev.for_metric(m, eddington_ratio=0.2)
    r_e = 5.469668466100368
    datum = TF.device_tracer(m, x, 2 * x[1], G.chart_for_metric(m, 2 * x[1]), G.ConstPointFunctions.redshift(m, x), ens)
    th = np.linspace(0.0, 2 * math.pi, 1441)[:-1]

    def area(r):
        h = float(d.cross_section(float(r)))
        rr = TF.find_offsets_for_radius(datum, np.full(th.size, r), th, r_min=m.inner_radius(), β0=2.0, heights=np.full(th.size, h))[0]
        return 0.5 * np.sum(rr * rr) * (th[1] - th[0])

    dA = (area(r_e + 1e-3) - area(r_e - 1e-3)) / 2e-3
    for N, rel in ((80, 5e-4), (400, 1e-4)):
        c = G.cunningham_transfer_function(m, x, d, r_e, β0=2.0, ensemble=ens, N=N)
        assert np.all(np.isfinite(c.f))
        gs = np.clip(c.g_star, 0.0, 1.0)
        y = c.f / (c.gmin + gs * (c.gmax - c.gmin))
        for k in np.flatnonzero(c.f == 0.0):
            y[k] = 0.5 * (y[k - 1] + y[(k + 1) % y.size])
        phi = np.arcsin(np.sqrt(gs))
        total = np.sum(0.5 * (y + np.roll(y, -1)) * 2.0 * np.abs(np.roll(phi, -1) - phi))
        assert total == pytest.approx(dA / (math.pi * r_e), rel=rel), N
        if N == 80:
            assert float(np.nansum(c.f)) == pytest.approx(21.4028, abs=2e-3)
