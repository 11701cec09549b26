"""Pin the CPU oracle against every golden value the reference's own tests hold for the render
path plus closed-form checks (K7, K8)."""
import math

import numpy as np
import pytest

X_SMOKE = np.array([0.0, 100.0, math.radians(85), 0.0])


def _fingerprint(O, metric, params, disc):
    # _run_rendergeodesics, test/smoke-tests/rendergeodesics.jl:16-30
    cfg = O.make_config(metric, params, disc=disc, lambda_max=200.0)
    img = O.rendergeodesics(cfg, X_SMOKE, (-9.5, 9.5), (-9.5, 9.5), 20, 20)
    return float(np.nansum(img))


# test/smoke-tests/rendergeodesics.jl:43-67.  The reference asserts rtol=1e-1 on values recorded
# to 16 digits; its own Kerr(a=0) vs Johannsen(0) values (the same spacetime) differ by 4.4e-7,
# so 1e-6 is the resolution of the golden data itself.
@pytest.mark.parametrize(
    "metric,params,disc,expected",
    [
        ("kerr", (1.0, 0.0), None, 9009.452876609641),                         # G1
        ("johannsen", (1.0, 0.0, 0.0, 0.0, 0.0, 0.0), None, 9009.448935932085),  # G2
        ("kerr", (1.0, 0.0), (0.0, 40.0), 38412.08347901267),                  # G3
        ("johannsen", (1.0, 0.0, 0.0, 0.0, 0.0, 0.0), (0.0, 40.0), 38412.08386562321),  # G4
    ],
)
def test_rendergeodesics_fingerprints(oracle, metric, params, disc, expected):
    got = _fingerprint(oracle, metric, params, disc)
    assert got == pytest.approx(expected, rel=1e-6)


# the other AbstractStaticAxisSymmetric metrics of the same reference test (rendergeodesics.jl:32-67);
# Morris-Thorne's shadow is the sum over a handful of rays that cross the throat (l <= 0) where steps
# are large, so its recorded value is only good to the size of one step (3e-5)
@pytest.mark.parametrize(
    "metric,params,disc,expected,rtol",
    [
        ("morris-thorne", (1.0,), None, 402.17907632733284, 1e-4),
        ("bumblebee", (1.0, 0.0, 0.0), None, 9009.452384885506, 1e-6),
        ("kerr-newman", (1.0, 0.0, 0.0), None, 9009.451384824908, 1e-6),
        ("morris-thorne", (1.0,), (0.0, 40.0), 9375.430228131403, 1e-6),
        ("bumblebee", (1.0, 0.0, 0.0), (0.0, 40.0), 38412.0832157869, 1e-6),
        ("kerr-newman", (1.0, 0.0, 0.0), (0.0, 40.0), 38412.08517225652, 1e-6),
    ],
)
def test_rendergeodesics_fingerprints_other_metrics(oracle, metric, params, disc, expected, rtol):
    assert _fingerprint(oracle, metric, params, disc) == pytest.approx(expected, rel=rtol)


def test_johannsen_psaltis_chart_and_kerr_newman_fingerprints(oracle):
    u = np.array([0.0, 1000.0, math.pi / 2, 0.0])
    # test/integration/test-charts.jl:5-18 (rtol 1e-4 there)
    cfg = oracle.make_config("johannsen-psaltis", (1.0, 0.8831, 0.4), lambda_max=2000.0)
    img = oracle.rendergeodesics(cfg, u, (-8, 8), (-8, 8), 100, 100)
    assert float(np.nansum(img)) == pytest.approx(2.9619136946153212e6, rel=1e-6)
    # test/unit/metrics.kerr-newman.jl:7-25, q = 0 (rtol 1e-3 there)
    cfg = oracle.make_config("kerr-newman", (1.0, 0.6, 0.6), lambda_max=2000.0)
    img = oracle.rendergeodesics(cfg, u, (-8, 8), (-8, 8), 40, 40)
    assert float(np.nansum(img)) == pytest.approx(428809.9681726607, rel=1e-6)
    # PoloidalShapeChart from the event-horizon shape, test/integration/test-charts.jl:20-31 (rtol 1e-4 there);
    # the table is solved here independently of the package (dense scan of g_tϕ² - g_tt g_ϕϕ in r)
    ths = np.linspace(1e-7, 2 * math.pi - 1e-7, 100)
    rgrid = np.linspace(0.0, 5.0, 500001)[1:]
    tab = np.empty(100)
    for i, th in enumerate(ths):
        M_, a_, e3 = 1.0, 0.8831, 0.4
        s2, c2 = math.sin(th) ** 2, math.cos(th) ** 2
        Sig = rgrid ** 2 + a_ * a_ * c2
        h = e3 * M_ ** 3 * rgrid / Sig ** 2
        tt = -(1 + h) * (1 - 2 * M_ * rgrid / Sig)
        pp = s2 * (rgrid ** 2 + a_ * a_ + 2 * a_ * a_ * M_ * rgrid * s2 / Sig) + h * a_ * a_ * (Sig + 2 * M_ * rgrid) * s2 * s2 / Sig
        tp = -a_ * 2 * M_ * rgrid * s2 * (1 + h) / Sig
        f = tp * tp - tt * pp
        k = np.nonzero(np.signbit(f[:-1]) != np.signbit(f[1:]))[0][-1]
        tab[i] = rgrid[k] - f[k] * (rgrid[k + 1] - rgrid[k]) / (f[k + 1] - f[k])
    cfg = oracle.make_config("johannsen-psaltis", (1.0, 0.8831, 0.4), lambda_max=2000.0, chart_table=tab * 1.001,
                             chart_theta=(ths[0], ths[-1]))
    img = oracle.rendergeodesics(cfg, u, (-8, 8), (-8, 8), 100, 100)
    assert float(np.nansum(img)) == pytest.approx(2.9540649115176247e6, rel=1e-7)
    # charged test particles: Lorentz force from the Faraday tensor, same file :26-27 (rtol 1e-3 there)
    for q, gold in ((1.0, 253280.6794972752), (-1.0, 619335.5670363897)):
        cfg = oracle.make_config("kerr-newman", (1.0, 0.6, 0.6), lambda_max=2000.0, q=q)
        img = oracle.rendergeodesics(cfg, u, (-8, 8), (-8, 8), 40, 40)
        assert float(np.nansum(img)) == pytest.approx(gold, rel=1e-7)


def _count_inner(O, G, plane):
    # test/image-planes/test-polar-grids.jl:9-21, test/utils.jl:1-4
    m = G.KerrMetric()
    u = np.array([1.0, 1e3, math.pi / 2, 0.0])
    a, b = G.impact_parameters(plane, u)
    cfg = O.make_config("kerr", (1.0, 0.0), lambda_max=2000.0)
    v = O.map_impact_parameters(cfg, u, a, b)
    pts = O.trace(cfg, u, v)
    return int(np.sum(pts["status"] == O.WITHIN_INNER_BOUNDARY))


def test_polar_grid_inner_boundary_counts(oracle, G):   # G5
    assert _count_inner(oracle, G, G.PolarPlane(G.LinearGrid(), Nr=10, Nθ=10)) == 10
    assert _count_inner(oracle, G, G.PolarPlane(G.GeometricGrid(), Nr=10, Nθ=10)) == 30
    assert _count_inner(oracle, G, G.PolarPlane(G.InverseGrid(), Nr=10, Nθ=10)) == 80


def test_cartesian_grid_inner_boundary_counts(oracle, G):   # G5
    kw = dict(x_min=0.1, y_min=0.1, Nx=12, Ny=12)
    assert _count_inner(oracle, G, G.CartesianPlane(G.LinearGrid(), **kw)) == 1
    assert _count_inner(oracle, G, G.CartesianPlane(G.GeometricGrid(), **kw)) == 25
    assert _count_inner(oracle, G, G.CartesianPlane(G.InverseGrid(), **kw)) == 81


def test_tsit5_tableau_identities(oracle):
    # SURVEY App. A.1/A.2 consistency identities
    c, a, bt, r = oracle.tsit5_tableau()
    for i in range(7):
        assert a[i].sum() == pytest.approx(c[i], abs=3e-15)
    assert a[6].sum() == pytest.approx(1.0, abs=3e-15)
    assert bt.sum() == pytest.approx(0.0, abs=3e-15)
    for k in range(1, 5):
        assert np.dot(a[6], c ** k) == pytest.approx(1.0 / (k + 1), abs=3e-15)
    b_at_1 = r.sum(axis=1)
    np.testing.assert_allclose(b_at_1[:6], a[6, :6], atol=3e-15)
    assert b_at_1[6] == pytest.approx(0.0, abs=3e-15)


def _kerr_lnrbasis(M, a, r, th):
    # test/unit/orthonormalization.jl:84-97
    Sig = r * r + (a * math.cos(th)) ** 2
    Del = r * r - 2 * M * r + a * a
    A = (r * r + a * a) ** 2 - a * a * Del * math.sin(th) ** 2
    om = 2 * M * a * r / A
    cols = [
        math.sqrt(Sig * Del / A) * np.array([1.0, 0, 0, 0]),
        math.sqrt(Sig / Del) * np.array([0, 1.0, 0, 0]),
        math.sqrt(Sig) * np.array([0, 0, 1.0, 0]),
        math.sqrt(A / Sig) * math.sin(th) * np.array([-om, 0, 0, 1.0]),
    ]
    return np.column_stack(cols)


def _kerr_lnrframe(M, a, r, th):
    # test/unit/orthonormalization.jl:52-65
    Sig = r * r + (a * math.cos(th)) ** 2
    Del = r * r - 2 * M * r + a * a
    A = (r * r + a * a) ** 2 - a * a * Del * math.sin(th) ** 2
    om = 2 * M * a * r / A
    cols = [
        math.sqrt(A / (Sig * Del)) * np.array([1.0, 0, 0, om]),
        math.sqrt(Del / Sig) * np.array([0, 1.0, 0, 0]),
        math.sqrt(1 / Sig) * np.array([0, 0, 1.0, 0]),
        math.sqrt(Sig / A) / math.sin(th) * np.array([0, 0, 0, 1.0]),
    ]
    return np.column_stack(cols)


ANGLES = [0.3, 0.9, math.pi / 2, 2.1, 2.9]


def test_lnr_tetrads_match_analytic_kerr(oracle):   # K1
    for M in (0.2, 1.0, 1.8):
        for a in np.arange(-M, M + 1e-9, 0.5):
            a = float(a)
            cfg = oracle.make_config("kerr", (M, a))
            rin = M + math.sqrt(M * M - a * a)
            for th in ANGLES:
                x = np.array([0.0, rin + 0.3, th, 0.1])
                np.testing.assert_allclose(oracle.lnrbasis(cfg, x), _kerr_lnrbasis(M, a, x[1], th), atol=1e-10)
                x = np.array([0.0, rin + 4.2, th, 0.0])
                np.testing.assert_allclose(oracle.lnrframe(cfg, x), _kerr_lnrframe(M, a, x[1], th), atol=1e-13)


def test_lnrframe_gives_minkowski(oracle):   # K2
    for metric, params in (("kerr", (1.0, 0.6)), ("johannsen", (1.0, 0.6, 1.0, 0.5, 0.0, 0.3))):
        cfg = oracle.make_config(metric, params)
        for th in ANGLES:
            x = np.array([0.0, 7.3, th, 0.0])
            F = oracle.lnrframe(cfg, x)
            g, _, _ = oracle.metric_jacobian(cfg, x[1], x[2])
            Gm = np.diag(g[:4]).copy()
            Gm[0, 3] = Gm[3, 0] = g[4]
            np.testing.assert_allclose(F.T @ Gm @ F, np.diag([-1.0, 1, 1, 1]), atol=1e-12)


def test_isco_table(oracle):   # K3: test/smoke-tests/special-radii.jl:20-38, test/test-special-radii.jl:7-10
    def isco(metric, params):
        return oracle.isco(oracle.make_config(metric, params))

    assert isco("kerr", (1.0, 0.0)) == 6.0
    assert isco("kerr", (1.0, 1.0)) == 1.0
    assert isco("kerr", (1.0, 0.998)) == pytest.approx(1.2369706551751847, abs=1e-12)
    assert isco("kerr", (1.0, -0.998)) == pytest.approx(8.99437445480357, abs=1e-12)
    assert isco("johannsen", (1.0, 0.0, 0, 0, 0, 0)) == pytest.approx(6.0, abs=1e-5)
    assert isco("johannsen", (1.0, 0.998, 0, 0, 0, 0)) == pytest.approx(1.2369706551751847, abs=1e-5)
    assert isco("johannsen", (1.0, 0.998, 1.0, 0, 0, 0)) == pytest.approx(2.8482863127671534, abs=1e-5)
    assert isco("johannsen", (1.0, 0.998, 0.0, 1.0, 0, 0)) == pytest.approx(1.1306596884484472, abs=1e-5)


def test_keplerian_vphi_sums(oracle):   # K4: test/smoke-tests/circular-orbits.jl:11-22 (atol 1e-6)
    rs = np.arange(6.0, 10.0 + 1e-9, 0.5)

    def total(metric, params):
        cfg = oracle.make_config(metric, params)
        return sum(oracle.circular_fourvelocity(cfg, r)[3] for r in rs)

    assert total("kerr", (1.0, 0.0)) == pytest.approx(0.5432533297869712, abs=1e-6)
    assert total("kerr", (1.0, 1.0)) == pytest.approx(0.5016710246454921, abs=1e-6)
    assert total("kerr", (1.0, -1.0)) == pytest.approx(0.5993458160081419, abs=2e-6)
    assert total("johannsen", (1.0, 1.0, 0.0, 1.0, 0.0, 0.0)) == pytest.approx(0.4980454719932759, abs=1e-6)


def _conserved(O, cfg, x, v):
    g, _, _ = O.metric_jacobian(cfg, x[1], x[2])
    E = -(g[0] * v[0] + g[4] * v[3])
    L = g[4] * v[0] + g[3] * v[3]
    norm = g[0] * v[0] ** 2 + g[1] * v[1] ** 2 + g[2] * v[2] ** 2 + g[3] * v[3] ** 2 + 2 * g[4] * v[0] * v[3]
    return E, L, norm


def test_conservation_and_closed_form_redshift(oracle):   # K7 + K8
    M, a = 1.0, 0.998
    isco = oracle.isco(oracle.make_config("kerr", (M, a)))
    cfg = oracle.make_config("kerr", (M, a), disc=(isco, 50.0), lambda_max=2000.0)
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    W = H = 24
    img, pts = oracle.rendergeodesics(cfg, x, (-60, 60), (-35, 35), W, H, pf_id=oracle.PF_REDSHIFT,
                                      filter_id=oracle.FILTER_INTERSECTED, r_isco=isco, return_points=True)
    hits = pts[pts["status"] == oracle.INTERSECTED_WITH_GEOMETRY]
    assert len(hits) > 20
    g_img = img.T.ravel()[pts["status"] == oracle.INTERSECTED_WITH_GEOMETRY]
    on_surface = 0
    for gp, gval in zip(hits, g_img):
        E0, L0, n0 = _conserved(oracle, cfg, gp["x_init"], gp["v_init"])
        E1, L1, n1 = _conserved(oracle, cfg, gp["x"], gp["v"])
        assert E1 == pytest.approx(E0, rel=2e-7)
        assert L1 == pytest.approx(L0, rel=2e-7, abs=1e-7)
        assert abs(n1) < 1e-6
        # hit surface is |cosθ| = gtol, except rays that enter through the disc's radial edge
        assert abs(math.cos(gp["x"][2])) <= 0.01 + 1e-9
        on_surface += abs(abs(math.cos(gp["x"][2])) - 0.01) < 1e-9
        rho = gp["x"][1] * abs(math.sin(gp["x"][2]))
        if rho >= isco:
            # regular_pdotu_inv (src/redshift.jl:166-167): g = 1/(u^t (1 - Ω L/E))
            vd = oracle.circular_fourvelocity(cfg, rho)
            Om = vd[3] / vd[0]
            # exact only on the equator; the end point sits at |cosθ| = 0.01
            assert gval == pytest.approx(1.0 / (vd[0] * (1.0 - Om * L1 / E1)), rel=5e-3)
    assert on_surface >= 0.9 * len(hits)
