"""GR_METRIC_TABULATED: a user-defined AbstractStaticAxisSymmetric metric on the device.

The reference's plugin contract is one method, metric_components(m, (r, θ)) (src/metrics/kerr-metric.jl:62-70), with ForwardDiff
for the Jacobian (auto-diff.jl:206-211).  Here the host samples that method, gr_metric_table_fit turns the samples into piecewise
polynomials and the kernels evaluate components and derivatives from the table.  Checked:

  CPU  the table against the oracle's dual-number Jacobian (Kerr, Johannsen, a metric of no catalogue), the fold in θ, the
       error paths of the four host-only entry points, and the device functor compiled for the host (tests/host_harness.cpp)
       against the oracle ray by ray;
  GPU  tabulated-Kerr on C2 and tabulated-Johannsen on C4, every pixel, against the fused kernels AND the oracle at the bar of
       tests/test_gpu_baseline_configs.py; the metric of no catalogue against the oracle running the same function through its
       dual numbers.
"""
import ctypes as C
import math

import numpy as np
import pytest

import harness as Hh

X_FAR = np.array([0.0, 1000.0, math.radians(75), 0.0])
BUMP = (1.0, 0.9, 0.08, 6.0, 2.0)       # M, a, ϵ, r_b, w of oracle/metrics_tmpl.h test_bump_components


def bump_components(r, th, p=BUMP):
    """The same function as the oracle's stand-in for a user-defined metric, as a numpy callable: Kerr with
    g_tt scaled by 1 + ϵ sin²θ / (1 + ((r - r_b)/w)²)."""
    M, a, e, rb, w = p
    s2 = np.sin(th) ** 2
    c2 = 1.0 - s2
    Sig = r * r + a * a * c2
    tt = -(1.0 - 2.0 * M * r / Sig)
    rr = Sig / (r * r + a * a - 2.0 * M * r)
    pp = s2 * (r * r + a * a + 2.0 * M * r * a * a * s2 / Sig)
    tp = -2.0 * M * r * a * s2 / Sig
    x = (r - rb) / w
    return (tt * (1.0 + e * s2 / (1.0 + x * x)), rr, Sig, pp, tp)


def _oracle_cfg(O, name, params):
    return O.make_config(name, params)


def _jacobian_errors(O, tm, cfg, rng, n=400):
    """Largest errors of the table's (g, ∂r g, ∂θ g) against the oracle's dual numbers at random points of the table's range.
    Each error is taken point by point relative to the component's size THERE -- with floors where a component passes through
    zero without the geodesic equation caring: g_tt on the ergosurface (against 0.1), g_ϕϕ at the poles (against 1e-3 g_θθ),
    g_tϕ (against 1e-2 of the block it couples) -- and the radial derivative per unit of ln(r - r0), as the fit quotes it."""
    rs = tm.r_min * (tm.r_max / tm.r_min) ** rng.uniform(0, 1, n)
    ths = rng.uniform(0.0, math.pi, n)
    A, B = [], []
    for r, th in zip(rs, ths):
        A.append(np.concatenate(tm.table_jacobian(r, th)))
        B.append(np.concatenate(O.metric_jacobian(cfg, r, th)))
    A, B = np.array(A), np.array(B)
    sc = np.abs(B[:, :5]).copy()
    sc[:, 0] = np.maximum(sc[:, 0], 0.1)
    sc[:, 3] = np.maximum(sc[:, 3], 1e-3 * sc[:, 2])
    sc[:, 4] = np.maximum(sc[:, 4], 1e-2 * np.sqrt(sc[:, 0] * sc[:, 3]))
    ev = float(np.max(np.abs(A[:, :5] - B[:, :5]) / sc))
    edr = float(np.max(np.abs(A[:, 5:10] - B[:, 5:10]) * (rs - tm.r0)[:, None] / sc))
    edt = float(np.max(np.abs(A[:, 10:] - B[:, 10:]) / sc))
    return ev, edr, edt


@pytest.fixture(scope="module")
def tab_kerr(G):
    return G.TabulatedMetric(G.KerrMetric(1.0, 0.998))


@pytest.fixture(scope="module")
def tab_johannsen(G):
    return G.TabulatedMetric(G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0))


@pytest.fixture(scope="module")
def tab_bump(G, oracle):
    # the oracle's stand-in for a user-defined metric lives in a library of its own (oracle/Makefile `usermetric`)
    with oracle.user_metric_library():
        isco = oracle.isco(oracle.make_config("test-bump", BUMP))
    return G.TabulatedMetric(bump_components, inner_radius=1.0 + math.sqrt(1.0 - 0.81), isco=isco)


def _oracle_lib(oracle, name):
    """the oracle library that knows metric `name`"""
    import contextlib

    return oracle.user_metric_library() if name == "test-bump" else contextlib.nullcontext()


def test_table_matches_dual_number_jacobian_kerr(G, oracle, tab_kerr):
    assert tab_kerr.m_r == 24 and tab_kerr.n_theta == 96          # the default grid suffices: no refinement
    assert tab_kerr.errors[0] < 1e-10 and tab_kerr.errors[1] < 1e-7 and tab_kerr.errors[2] < 1e-7
    cfg = oracle.make_config("kerr", (1.0, 0.998))
    ev, edr, edt = _jacobian_errors(oracle, tab_kerr, cfg, np.random.default_rng(5))
    assert ev < 3e-11 and edr < 3e-8 and edt < 3e-8, (ev, edr, edt)


def test_table_matches_dual_number_jacobian_johannsen(G, oracle, tab_johannsen):
    cfg = oracle.make_config("johannsen", (1.0, 0.7, 2.0, 0.0, 0.0, 1.0))
    ev, edr, edt = _jacobian_errors(oracle, tab_johannsen, cfg, np.random.default_rng(6))
    assert ev < 3e-11 and edr < 3e-8 and edt < 3e-8, (ev, edr, edt)


def test_table_matches_dual_number_jacobian_user_metric(G, oracle, tab_bump):
    cfg = oracle.make_config("test-bump", BUMP)
    with oracle.user_metric_library():
        ev, edr, edt = _jacobian_errors(oracle, tab_bump, cfg, np.random.default_rng(7))
    assert ev < 3e-11 and edr < 3e-8 and edt < 3e-8, (ev, edr, edt)


def test_error_estimates_are_not_optimistic(G, oracle):
    """The fit's own estimates (what a caller refines against) bound the measured errors on a deliberately coarse grid."""
    tm = G.TabulatedMetric(G.KerrMetric(1.0, 0.998), m_r=2, n_theta=8, max_refinements=0, strict=False)
    assert tm.m_r == 2 and tm.n_theta == 8
    cfg = oracle.make_config("kerr", (1.0, 0.998))
    ev, edr, edt = _jacobian_errors(oracle, tm, cfg, np.random.default_rng(8))
    assert tm.errors[0] > 1e-8                    # coarse on purpose
    assert ev < 3 * tm.errors[0] and edr < 3 * tm.errors[1] and edt < 3 * tm.errors[2], (ev, edr, edt, tm.errors)


def test_refinement_stops_at_tolerance(G):
    tm = G.TabulatedMetric(G.KerrMetric(1.0, 0.5), m_r=4, n_theta=8, max_refinements=8)
    assert tm.errors[0] <= 1e-10 and tm.errors[1] <= 1e-7 and tm.errors[2] <= 1e-7
    assert tm.m_r > 4 and tm.n_theta > 8


def test_theta_fold(tab_kerr):
    r, th = 3.7, 0.83
    g, dr, dt = tab_kerr.table_jacobian(r, th)
    for th2, sign in ((-th, -1.0), (th + 2 * math.pi, 1.0), (2 * math.pi - th, -1.0), (th - 4 * math.pi, 1.0)):
        g2, dr2, dt2 = tab_kerr.table_jacobian(r, th2)
        np.testing.assert_allclose(g2, g, rtol=1e-13)
        np.testing.assert_allclose(dr2, dr, rtol=1e-12)
        np.testing.assert_allclose(dt2, sign * dt, rtol=1e-11, atol=1e-13)
    # across the pole: θ = π + δ is θ = π - δ
    g3, _, dt3 = tab_kerr.table_jacobian(r, math.pi + 0.01)
    g4, _, dt4 = tab_kerr.table_jacobian(r, math.pi - 0.01)
    np.testing.assert_allclose(g3, g4, rtol=1e-13)
    np.testing.assert_allclose(dt3, -dt4, rtol=1e-11)


def test_outside_the_radial_range_takes_the_nearest_patch(tab_kerr):
    g, dr, dt = tab_kerr.table_jacobian(tab_kerr.r0 - 1.0, 1.0)      # inside the origin of the octaves: clamped, finite
    assert np.all(np.isfinite(g)) and np.all(np.isfinite(dr))
    g, dr, dt = tab_kerr.table_jacobian(float("nan"), 1.0)
    assert np.all(np.isfinite(g))


def test_host_entry_points_reject_bad_input(G):
    L = G._lib
    lib = L.load()
    grid = L.gr_metric_grid()
    assert lib.gr_metric_grid_plan(2.0, 1.0, 0.5, 8, 32, grid) == -1            # r_max < r_min
    assert lib.gr_metric_grid_plan(1.0, 10.0, 1.5, 8, 32, grid) == -1           # r0 >= r_min
    assert lib.gr_metric_grid_plan(1.0, 10.0, 0.5, 0, 32, grid) == -1
    assert lib.gr_metric_grid_plan(1.0, 10.0, 0.5, 8, 0, grid) == -1
    assert lib.gr_metric_grid_plan(1.0, 10.0, 0.5, 2, 2, None) == -1
    assert lib.gr_metric_grid_plan(1.0, 10.0, 0.5, 2, 2, grid) == 0
    assert grid.pole_factor == 1
    grid.pole_factor = 0                      # the samples below are constants: nothing vanishes on the axis
    assert grid.degree == 5 and grid.n_seg == 1 and grid.n_rows == grid.n_oct * 2
    assert grid.n_r_nodes == grid.n_oct * 2 * grid.fit_nodes and grid.n_theta_nodes == 2 * grid.fit_nodes
    rn, tn = np.empty(grid.n_r_nodes), np.empty(grid.n_theta_nodes)
    assert lib.gr_metric_grid_nodes(grid, rn.ctypes.data, tn.ctypes.data) == 0
    assert rn.min() > 1.0 - 0.5 and np.all(np.diff(np.sort(tn)) > 0) and tn.min() > 0 and tn.max() < math.pi
    samples = np.ones((grid.n_r_nodes, grid.n_theta_nodes, 5))
    table = np.empty(grid.table_doubles)
    err = (C.c_double * 3)()
    samples[3, 4, 2] = np.inf
    assert lib.gr_metric_table_fit(grid, samples.ctypes.data, table.ctypes.data, err) == -1
    assert b"finite" in lib.gr_last_error()
    samples[3, 4, 2] = 1.0
    bad = L.gr_metric_grid.from_buffer_copy(grid)
    bad.n_r_nodes += 1
    assert lib.gr_metric_table_fit(bad, samples.ctypes.data, table.ctypes.data, err) == -1
    assert lib.gr_metric_table_fit(grid, samples.ctypes.data, table.ctypes.data, err) == 0
    assert err[0] < 1e-14                     # a constant is fitted exactly
    g, dr, dt = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
    assert lib.gr_metric_table_eval(table.ctypes.data, table.size, 3.0, 1.0, g, dr, dt) == 0
    np.testing.assert_allclose(np.array(g), 1.0, atol=1e-14)
    np.testing.assert_allclose(np.array(dr), 0.0, atol=1e-13)
    assert lib.gr_metric_table_eval(table.ctypes.data, table.size - 1, 3.0, 1.0, g, dr, dt) == -1      # wrong length
    table[0] = 0.0
    assert lib.gr_metric_table_eval(table.ctypes.data, table.size, 3.0, 1.0, g, dr, dt) == -1          # not a table


def test_two_tables_have_distinct_build_ids(G):
    a = G.TabulatedMetric(G.KerrMetric(1.0, 0.5), m_r=2, n_theta=4, max_refinements=0, strict=False)
    b = G.TabulatedMetric(G.KerrMetric(1.0, 0.6), m_r=2, n_theta=4, max_refinements=0, strict=False)
    assert a.table[8] != b.table[8]           # H_BUILD_ID: the contexts' device copies are keyed by it


def test_scalar_callable_is_sampled_point_by_point(G):
    def f(r, th):
        if isinstance(r, np.ndarray):
            raise TypeError("scalars only")
        return G.KerrMetric(1.0, 0.3).metric_components(r, th)

    tm = G.TabulatedMetric(f, inner_radius=1.0 + math.sqrt(1 - 0.09), isco=5.0, m_r=2, n_theta=4, r_max=50.0, max_refinements=0, strict=False)
    g, _, _ = tm.table_jacobian(10.0, 1.0)
    np.testing.assert_allclose(g, f(10.0, 1.0), rtol=1e-6)


# ---- the device functor compiled for the host (tests/host_harness.cpp) against the oracle, ray by ray ----

def _compare_endpoints(got, ref, x_rtol=1e-6, max_flips=0, max_outliers=0, r_horizon=None):
    """status per ray, then positions / velocities / affine time at x_rtol (relative to the component, with a floor of 1e-3 of the
    state's largest) on all but `max_outliers` rays, which must still agree to 1e-3"""
    flips = int(np.sum(got["status"] != ref["status"]))
    assert flips <= max_flips, f"{flips} status flips"
    # rays that end on the inner boundary are compared by status only: t and ϕ diverge towards the horizon and the end state
    # there is ill-conditioned in every implementation (the fused Kerr kernel differs from the oracle by 2e-3 on the same rays)
    same = (got["status"] == ref["status"]) & (ref["status"] != 1)
    if r_horizon is not None:
        # ... and so are rays that run out of affine time while they hover just outside it (status NoStatus at r within 5 % of
        # the horizon: t and ϕ wind up there; 18 of 9216 rays of the a = 0.9 user metric, the same ones on the host build)
        same &= ref["x"][:, 1] > 1.05 * r_horizon
    worst = np.abs(got["lambda_max"][same] / ref["lambda_max"][same] - 1.0)
    for f in ("x", "v"):
        a, b = got[f][same], ref[f][same]
        scale = np.maximum(np.abs(b), 1e-3 * np.max(np.abs(b), axis=1, keepdims=True))
        worst = np.maximum(worst, np.max(np.abs(a - b) / scale, axis=1))
    assert int(np.sum(worst >= x_rtol)) <= max_outliers, (int(np.sum(worst >= x_rtol)), float(worst.max()))
    assert worst.max() < 1e-3


@pytest.mark.parametrize("which", ["kerr", "johannsen", "bump"])
def test_tabulated_endpoints_vs_oracle_kernel_logic(G, oracle, which, tab_kerr, tab_johannsen, tab_bump):
    tm, name, params = {"kerr": (tab_kerr, "kerr", (1.0, 0.998)),
                        "johannsen": (tab_johannsen, "johannsen", (1.0, 0.7, 2.0, 0.0, 0.0, 1.0)),
                        "bump": (tab_bump, "test-bump", BUMP)}[which]
    W = H = 24
    disc = (3.0, 400.0)      # (outer rim outside the field of view: rim pixels flip between any two implementations)
    cfg = G.render_configuration(tm, X_FAR, G.ThinDisc(*disc), 2000.0, image_width=W, image_height=H,
                                 alpha_lims=(-60, 60), beta_lims=(-35, 35))
    got = Hh.render_endpoints(G, cfg)
    ocfg = oracle.make_config(name, params, disc=disc, lambda_max=2000.0)
    assert ocfg.r_inner == pytest.approx(cfg.chart.inner_radius, rel=1e-14) and ocfg.r_outer == cfg.chart.outer_radius
    with _oracle_lib(oracle, name):
        vs = oracle.render_velocities(ocfg, X_FAR, (-60, 60), (-35, 35), W, H)
        ref = oracle.trace(ocfg, X_FAR, vs)
    _compare_endpoints(got, ref, max_flips=2)       # (the fused Johannsen kernel has the same two horizon-rim flips against the oracle)
    if which != "bump":
        # ... and against the metric's own fused right-hand side through the same integrator: the table's error alone
        fcfg = G.render_configuration(tm.source, X_FAR, G.ThinDisc(*disc), 2000.0, image_width=W, image_height=H,
                                      alpha_lims=(-60, 60), beta_lims=(-35, 35))
        _compare_endpoints(got, Hh.render_endpoints(G, fcfg), x_rtol=1e-7, max_flips=0)


@pytest.mark.parametrize("which", ["kerr-dark-matter", "kerr-refractive", "dilaton-axion"])
def test_segments_and_axis_terms_in_the_kernel_logic(G, which):
    """The device functor compiled for the host through tables with SEGMENTS (rays that cross the mass shell / the corona's edge on
    their way to the disc) and with AXIS TERMS (an axion charge, a plane seen from near the axis) against the same metric's own
    right-hand side through the same integrator: the table's error alone."""
    import warnings

    base = {"kerr-dark-matter": G.KerrDarkMatter(1.0, 0.6, 2.0, 8.0, 7.0), "kerr-refractive": G.KerrRefractive(1.0, 0.5, 1.1, 20.0),
            "dilaton-axion": G.DilatonAxion(1.0, 0.18627144681883856, -0.2796909434574291, 1.0823340240923809)}[which]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tm = G.TabulatedMetric(base)
    assert tm.grid.n_seg == {"kerr-dark-matter": 3, "kerr-refractive": 5, "dilaton-axion": 1}[which]
    assert tm.grid.pole_factor == (2 if which == "dilaton-axion" else 1)
    x = np.array([0.0, 864.3, math.radians(15.8 if which == "dilaton-axion" else 70.0), 0.0])
    W, H = (40, 12) if which == "dilaton-axion" else (20, 20)
    fov = 23.6 if which == "dilaton-axion" else 12.0
    disc = G.ThinDisc(5.635412874390816, 66.84467146396622) if which == "dilaton-axion" else G.ThinDisc(3.0, 40.0)
    kw = dict(image_width=W, image_height=H, alpha_lims=(-fov, fov), beta_lims=(-fov, fov))
    if which != "dilaton-axion":
        # A right-hand side with a kink (or a step) limits what tolerance 1e-9 resolves, whatever evaluates it: the fused form against
        # ITSELF at 1e-11 moves these rays by 2e-7 ... 2e-5 (the table against the converged trace: 4e-6).  At 1e-11 both are converged.
        kw.update(abstol=1e-11, reltol=1e-11)
    # (the table starts at the outermost horizon, the fused form at the reference's inner_radius: the same chart for both)
    chart = G.chart_for_metric(tm, 2000.0)
    got = Hh.render_endpoints(G, G.render_configuration(tm, x, disc, 2000.0, chart=chart, **kw))
    ref = Hh.render_endpoints(G, G.render_configuration(base, x, disc, 2000.0, chart=chart, **kw))
    assert (ref["status"] == G.StatusCodes.IntersectedWithGeometry).sum() > 20
    _compare_endpoints(got, ref, x_rtol=1e-6, max_flips=2, max_outliers=2, r_horizon=tm.inner_radius())


# ---- on the device ----

def _render_pair(G, ens, tm, base, size, disc, pf_of):
    """(image through the table, image through the metric's own kernels) of one plane"""
    out = []
    for m in (tm, base):
        a, b, img = G.rendergeodesics(m, X_FAR, disc, 2000.0, image_width=size, image_height=size, alpha_lims=(-60, 60),
                                      beta_lims=(-35, 35), pf=pf_of(m), ensemble=ens)
        out.append(img)
    return out


def _oracle_strided(oracle, name, params, disc, size, stride, alims, blims, **pfkw):
    """The oracle's redshift image of every `stride`-th pixel (both directions) of the size² plane: pixel k of a range of `size`
    points is α0 + k Δ (rendering.jl:151-152), so the strided pixels are a plane of size / stride points from α0 to α0 + (size - stride) Δ."""
    n = size // stride
    cut = lambda lims: (lims[0], lims[0] + (size - stride) * (lims[1] - lims[0]) / (size - 1))
    cfg = oracle.make_config(name, params, disc=disc, lambda_max=2000.0)
    return oracle.rendergeodesics(cfg, X_FAR, cut(alims), cut(blims), n, n, pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_INTERSECTED, **pfkw)


@pytest.mark.gpu
def test_tabulated_kerr_equals_fused_kerr_and_oracle_on_c2(G, ens, oracle, tab_kerr):
    """BASELINE config C2 (Kerr a = 0.998, 1024², ThinDisc, redshift) through the TABLE: every pixel against the fused Kerr
    kernel, a strided subset against the oracle."""
    base = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(base.isco(), 50.0)
    pf_of = lambda m: G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    tab, fused = _render_pair(G, ens, tab_kerr, base, 1024, d, pf_of)
    flips = int(np.sum(np.isnan(tab) != np.isnan(fused)))
    both = ~np.isnan(tab) & ~np.isnan(fused)
    rel = np.abs(tab[both] / fused[both] - 1.0)
    # two integrations of the same rays with right-hand sides that differ by ~1e-10: the bar of the C2 test
    assert flips <= 40, flips
    assert np.quantile(rel, 0.999) < 1e-6 and np.median(rel) < 1e-8, (float(np.quantile(rel, 0.999)), float(np.median(rel)))
    assert int(np.sum(rel > 1e-6)) <= 200, int(np.sum(rel > 1e-6))
    # ... and every 8th pixel in both directions against the oracle (the plane's pixel (i, j) is pixel (i / 8, j / 8) of a 128² plane
    # whose limits are moved by half the difference of the pixel sizes: the same impact parameters)
    ref = _oracle_strided(oracle, "kerr", (1.0, 0.998), (base.isco(), 50.0), 1024, 8, (-60, 60), (-35, 35), r_isco=base.isco())
    sub = tab[::8, ::8]
    both = ~np.isnan(sub) & ~np.isnan(ref)
    assert both.sum() > 1000 and int(np.sum(np.isnan(sub) != np.isnan(ref))) <= 40
    rel_o = np.abs(sub[both] / ref[both] - 1.0)
    assert np.quantile(rel_o, 0.99) < 1e-6 and int(np.sum(rel_o > 1e-6)) <= 40, (float(np.quantile(rel_o, 0.99)), int(np.sum(rel_o > 1e-6)))


@pytest.mark.gpu
def test_tabulated_johannsen_equals_fused_johannsen_on_c4(G, ens, tab_johannsen):
    base = tab_johannsen.source
    d = G.ThinDisc(base.isco(), 50.0)
    pf_of = lambda m: G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    tab, fused = _render_pair(G, ens, tab_johannsen, base, 1024, d, pf_of)
    flips = int(np.sum(np.isnan(tab) != np.isnan(fused)))
    both = ~np.isnan(tab) & ~np.isnan(fused)
    rel = np.abs(tab[both] / fused[both] - 1.0)
    assert flips <= 40, flips
    assert np.quantile(rel, 0.999) < 1e-6 and np.median(rel) < 1e-8, (float(np.quantile(rel, 0.999)), float(np.median(rel)))
    assert int(np.sum(rel > 1e-6)) <= 200, int(np.sum(rel > 1e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["kerr", "johannsen", "bump"])
def test_tabulated_endpoints_vs_oracle_on_device(G, ens, oracle, which, tab_kerr, tab_johannsen, tab_bump):
    """End points of a 96 x 96 plane through the table on the DEVICE against the oracle's dual-number trace of the same
    function -- for the metric of no catalogue this is the only thing it can be compared with."""
    tm, name, params = {"kerr": (tab_kerr, "kerr", (1.0, 0.998)),
                        "johannsen": (tab_johannsen, "johannsen", (1.0, 0.7, 2.0, 0.0, 0.0, 1.0)),
                        "bump": (tab_bump, "test-bump", BUMP)}[which]
    W = H = 96
    disc = (3.0, 400.0)
    _, _, cache = G.prerendergeodesics(tm, X_FAR, G.ThinDisc(*disc), 2000.0, image_width=W, image_height=H, alpha_lims=(-60, 60),
                                       beta_lims=(-35, 35), ensemble=ens)
    got = np.ascontiguousarray(np.asarray(cache.points).T).reshape(-1)          # (height, width) -> ray order (column-major)
    ocfg = oracle.make_config(name, params, disc=disc, lambda_max=2000.0)
    with _oracle_lib(oracle, name):
        vs = oracle.render_velocities(ocfg, X_FAR, (-60, 60), (-35, 35), W, H)
        ref = oracle.trace(ocfg, X_FAR, vs)
    # 9216 rays through two independent integrators: the handful that skim the photon orbit amplify a last-digit difference
    # (3.7e-5 on one ray of the user metric; the fused Johannsen kernel against the oracle on C4: 4 pixels above 1e-7 of 4e5)
    _compare_endpoints(got, ref, max_flips=8, max_outliers=5, r_horizon=tm.inner_radius())


@pytest.mark.gpu
def test_user_metric_redshift_image_vs_oracle(G, ens, oracle, tab_bump):
    """The whole render path for a metric of no catalogue: generic ISCO, device-traced plunging table, redshift ∘ filter,
    against the oracle doing all of that with its dual numbers."""
    W = H = 64
    isco = tab_bump.isco()
    d = G.ThinDisc(isco, 40.0)
    pf = G.ConstPointFunctions.redshift(tab_bump, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    a, b, img = G.rendergeodesics(tab_bump, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=(-50, 50),
                                  beta_lims=(-30, 30), pf=pf, ensemble=ens)
    ocfg = oracle.make_config("test-bump", BUMP, disc=(isco, 40.0), lambda_max=2000.0)
    with oracle.user_metric_library():
        ref = oracle.rendergeodesics(ocfg, X_FAR, (-50, 50), (-30, 30), W, H, pf_id=oracle.PF_REDSHIFT,
                                     filter_id=oracle.FILTER_INTERSECTED, r_isco=isco)
    assert int(np.sum(np.isnan(img) != np.isnan(ref))) <= 4
    both = ~np.isnan(img) & ~np.isnan(ref)
    assert both.sum() > 500
    assert np.max(np.abs(img[both] / ref[both] - 1.0)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["kerr-dark-matter", "kerr-refractive"])
def test_piecewise_metrics_through_the_table_equal_their_fused_kernels(G, ens, which):
    """the two metrics of the catalogue that are piecewise in r, as CALLABLES with their break radii, through
    the table against their own fused kernels on 1024² -- the C2 scene and the C2 bar.  A right-hand side with a kink or a step
    bounds what a tolerance resolves whatever evaluates it (the fused kernel against itself at 1e-11 moves rays through the shell
    by up to 2e-5 at 1e-9, tests above), so both are traced at 1e-11, where both are converged."""
    ens.set("kernel", 2).set("precision", 64)
    base = {"kerr-dark-matter": G.KerrDarkMatter(1.0, 0.6, 2.0, 8.0, 7.0), "kerr-refractive": G.KerrRefractive(1.0, 0.5, 1.1, 20.0)}[which]
    # (circular orbits of this dark-matter shell have no ISCO -- dE/dr keeps its sign: the image is the affine time at the disc, which
    # feels every part of the ray's path; the refractive metric keeps Kerr's ISCO, kerr-refractive-ad.jl:61, and the redshift)
    isco = base.isco() if which == "kerr-refractive" else 6.0
    tm = G.TabulatedMetric(_as_callable(base), inner_radius=base.inner_radius(), isco=isco, breaks=base.break_radii())
    assert tm.grid.n_seg == (3 if which == "kerr-dark-matter" else 5) and (tm.m_r, tm.n_theta) == (24, 96)
    d = G.ThinDisc(isco, 50.0)
    out = []
    for m in (tm, base):
        pf = (G.ConstPointFunctions.redshift(m, X_FAR) if which == "kerr-refractive" else G.ConstPointFunctions.affine_time()) @ G.ConstPointFunctions.filter_intersected()
        _, _, img = G.rendergeodesics(m, X_FAR, d, 2000.0, image_width=1024, image_height=1024, alpha_lims=(-60, 60), beta_lims=(-35, 35),
                                      pf=pf, ensemble=ens, abstol=1e-11, reltol=1e-11, chart=G.chart_for_metric(tm, 12000.0))
        out.append(img)
    tab, fused = out
    flips = int(np.sum(np.isnan(tab) != np.isnan(fused)))
    both = ~np.isnan(tab) & ~np.isnan(fused)
    assert both.sum() > 250_000
    rel = np.abs(tab[both] / fused[both] - 1.0)
    print(f"  {which}: flips {flips}, median {np.median(rel):.2e}, q999 {np.quantile(rel, 0.999):.2e}, max {rel.max():.2e}, > 1e-6: {int(np.sum(rel > 1e-6))}")
    assert flips <= 60, flips
    assert np.quantile(rel, 0.999) < 1e-6 and np.median(rel) < 1e-8, (float(np.quantile(rel, 0.999)), float(np.median(rel)))
    assert int(np.sum(rel > 1e-6)) <= 200, int(np.sum(rel > 1e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(78, 16), (312, 64)])
def test_axis_grazing_rays_of_an_axion_charged_metric(G, ens, size):
    """The failing scene of round 5's soak (profiles/r5z_soak_tab_500_seed77.log:660, scene 405): a dilaton-axion metric with β != 0
    seen from 16° -- four rays that graze the polar axis ended 1e-4 ... 3e-2 off in ϕ through a table that held g_ϕϕ as sampled
    (an absolute 5e-12 next to its zero).  With the axis terms taken out (pole_factor 2) every ray meets the fused kernel's; the
    same plane four times finer puts sixteen times as many rays next to the axis."""
    import warnings

    ens.set("kernel", 2).set("precision", 64)
    base = G.DilatonAxion(1.0, 0.18627144681883856, -0.2796909434574291, 1.0823340240923809)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tm = G.TabulatedMetric(base)
    assert tm.grid.pole_factor == 2 and not any("milliradians" in str(x.message) for x in w)
    x = np.array([0.0, 864.3, math.radians(15.8), 0.0])
    d = G.ThinDisc(5.635412874390816, 66.84467146396622)
    kw = dict(image_width=size[0], image_height=size[1], alpha_lims=(-23.6, 23.6), beta_lims=(-23.6, 23.6), abstol=5.5e-10, reltol=5.5e-10,
              ensemble=ens, chart=G.chart_for_metric(tm, 2 * 864.3))
    ref = G.prerendergeodesics(base, x, d, 2 * 864.3, **kw)[2].points.ravel()
    got = G.prerendergeodesics(tm, x, d, 2 * 864.3, **kw)[2].points.ravel()
    # rays that pass within 5 mrad of the axis somewhere are what the scene is about: there are some
    _compare_endpoints(got, ref, x_rtol=1e-6, max_flips=2, max_outliers=0, r_horizon=tm.inner_radius())


@pytest.mark.gpu
def test_tangents_through_the_table_equal_the_tangent_oracle_ray_by_ray(G, ens, oracle, tab_kerr):
    """gr_ray_tangent of a TABULATED metric -- value + ∂/∂α + ∂/∂β ride through the table's own polynomials
    (the reference pushes Duals through a user's metric_components, precision-solvers.jl:401-451) -- pinned per ray on the oracle's
    dual-number trace of the Kerr metric, at the bar of the fused tangent kernel (tests/test_gpu_tangent.py), and against that
    kernel itself."""
    from gradus_jl_amd.transfer_functions import device_tracer

    ens.set("kernel", 2).set("precision", 64)
    a = 0.998
    kerr = G.KerrMetric(1.0, a)
    x = np.array([0.0, 100_000.0, math.radians(30), 0.0])
    rng = np.random.default_rng(2026)
    rr, th = rng.uniform(2.5, 14.0, 300), rng.uniform(0.0, 2 * math.pi, 300)
    th[:4] = [math.pi / 2, math.pi / 2 + 1e-3, 3 * math.pi / 2, 0.0]         # through / next to the polar axis, and the α axis
    al, be = rr * np.cos(th), rr * np.sin(th)
    dev = {}
    for name, m in (("table", tab_kerr), ("fused", kerr)):
        chart = G.chart_for_metric(m, 2 * x[1], closest_approach=1.005)
        pf = G.ConstPointFunctions.redshift(m, x, ensemble=ens)
        tr = device_tracer(m, x, 2 * x[1], chart, pf, ens)
        assert tr.tangent is not None
        dev[name] = tr.tangent(al, be)
    cfg = oracle.make_config("kerr", (1.0, a), disc={"datum": 0.0}, lambda_max=2 * x[1], closest_approach=1.005, outer_radius=2 * x[1])
    orc = oracle.ray_tangent(cfg, x, al, be, r_isco=kerr.isco(), max_time=2 * x[1], norm_with_tangents=True)

    def rel(d_, o_, cols):
        out = np.zeros(d_.shape[0])
        for lo in cols:
            scale = np.maximum(np.abs(o_[:, lo]), np.abs(o_[:, lo + 1]))
            out = np.maximum(out, np.max(np.abs(d_[:, lo:lo + 2] - o_[:, lo:lo + 2]), axis=1) / scale)
        return out

    t = dev["table"]
    assert np.array_equal(t[:, 7], orc[:, 7]) and np.array_equal(t[:, 7], dev["fused"][:, 7])
    hit = t[:, 7] == 2
    assert hit.sum() > 250
    # Where a ray meets the disc INSIDE the ISCO, g is measured against the plunging flow: Kerr's closed form in the fused kernel and
    # the oracle (redshift.jl:93-164), the traced and interpolated plunge for every other metric -- a tabulated one included
    # (redshift.jl:246-276): 6e-5 apart on the two such rays here.  The radius and its derivatives do not know about the disc's flow.
    outside = hit & (t[:, 1] > 1.02 * kerr.isco())
    assert 0 < (hit & ~outside).sum() < 10
    np.testing.assert_allclose(t[hit, 1], orc[hit, 1], rtol=1e-5)
    np.testing.assert_allclose(t[outside, 0], orc[outside, 0], rtol=1e-5)
    np.testing.assert_allclose(t[hit, 0], orc[hit, 0], rtol=1e-3)
    e_orc = np.maximum(rel(t[outside], orc[outside], (2, 4)), 0.0)
    e_fused = rel(t[outside], dev["fused"][outside], (2, 4))
    e_rad = rel(t[hit], orc[hit], (4,))
    print(f"  tangents through the table: vs the tangent oracle max {e_orc.max():.2e} median {np.median(e_orc):.2e} (∂ρ of all hits: {e_rad.max():.2e}); "
          f"vs the fused tangent kernel max {e_fused.max():.2e}")
    assert e_orc.max() < 1e-5 and np.median(e_orc) < 2e-6 and e_rad.max() < 1e-5
    assert e_fused.max() < 1e-6


@pytest.mark.gpu
def test_the_library_refuses_a_chart_or_an_observer_outside_the_table(G, ens):
    """nothing used to compare the radii a trace visits with the table's range -- beyond it the polynomials
    extrapolate to garbage (g_tt = -3e4 at ten times r_max) with rc = 0.  The C ABI now refuses; the Python host grows the table."""
    ens.set("kernel", 2).set("precision", 64)
    tm = G.TabulatedMetric(G.KerrMetric(1.0, 0.5), r_max=300.0)
    x = np.array([0.0, 200.0, 1.2, 0.0])
    cfg = G.render_configuration(tm, x, G.ThinDisc(3.0, 30.0), 500.0, image_width=8, image_height=8, alpha_lims=(-5, 5), beta_lims=(-5, 5),
                                 chart=G.chart_for_metric(tm, 290.0), ensemble=ens)
    c = cfg.abi_config()
    assert tm.r_max == 300.0                                     # the chart fits: the table is used as it is
    # at the C ABI: the same call with the chart's outer radius beyond the table, then with the observer beyond it
    L = G._lib.load()
    plane = cfg.abi_plane()
    rg = G._lib.gr_range(0, 64, 64, 1)
    pts = np.zeros(64, dtype=G._lib.POINT_DTYPE)

    ok = L.gr_render_endpoints(ens.ctx.handle, C.byref(c), C.byref(plane), C.byref(rg), pts.ctypes.data, None)
    assert ok == 0
    c.r_outer = 400.0
    assert L.gr_render_endpoints(ens.ctx.handle, C.byref(c), C.byref(plane), C.byref(rg), pts.ctypes.data, None) == -1
    assert b"leaves the radial range" in L.gr_last_error()
    c.r_outer = 290.0
    c.r_inner = 0.9 * tm.r_min
    assert L.gr_render_endpoints(ens.ctx.handle, C.byref(c), C.byref(plane), C.byref(rg), pts.ctypes.data, None) == -1
    c.r_inner = float(cfg.chart.inner_radius)
    plane.x_obs[1] = 350.0
    assert L.gr_render_endpoints(ens.ctx.handle, C.byref(c), C.byref(plane), C.byref(rg), pts.ctypes.data, None) == -1
    assert b"rays start at" in L.gr_last_error()
    # the Python host: an observer at 7000 makes the table grow (it used to trace through extrapolated polynomials)
    x2 = np.array([0.0, 7000.0, 1.2, 0.0])
    a_, b_, img = G.rendergeodesics(tm, x2, G.ThinDisc(3.0, 30.0), 15000.0, image_width=16, image_height=16, alpha_lims=(-20, 20), beta_lims=(-20, 20),
                                    ensemble=ens)
    assert tm.r_max >= 14000.0
    a_, b_, ref = G.rendergeodesics(tm.source, x2, G.ThinDisc(3.0, 30.0), 15000.0, image_width=16, image_height=16, alpha_lims=(-20, 20),
                                    beta_lims=(-20, 20), ensemble=ens)
    np.testing.assert_allclose(img, ref, rtol=1e-7)


@pytest.mark.gpu
def test_tabulated_metric_through_the_fp32_kernels(G, ens, tab_kerr):
    """`precision` 32 (the tolerance sweeps of BASELINE config 5) on a user's metric: the fp32 kernels evaluate the fp64 table in
    double and run inverse, contraction and step in single precision.  Against the fused Kerr kernels at the same precision and
    tolerance (what differs is rounding), against fp64 at 1e-9 at the bars of test_fp32_kernels_track_fp64, and a line profile
    through both."""
    base = tab_kerr.source
    d = G.ThinDisc(base.isco(), 50.0)
    kw = dict(image_width=256, image_height=256, alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens)
    imgs = {}
    _, _, ref64 = G.rendergeodesics(base, X_FAR, d, 2000.0, pf=G.ConstPointFunctions.redshift(base, X_FAR) @ G.ConstPointFunctions.filter_intersected(), **kw)
    ens.set("precision", 32)
    try:
        for name, m in (("tab", tab_kerr), ("fused", base)):
            pf = G.ConstPointFunctions.redshift(m, X_FAR, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
            _, _, img, st = G.rendergeodesics(m, X_FAR, d, 2000.0, pf=pf, abstol=1e-5, reltol=1e-5, stats=True, **kw)
            assert st["rays"] == 256 * 256 and st["flagged_rays"] <= 0.01 * st["rays"]
            imgs[name] = img
        plane = G.PolarPlane(G.GeometricGrid(), Nr=256, Nθ=256, r_min=1.0, r_max=60.0)
        u = np.array([0.0, 1000.0, math.radians(60), 0.0])
        bins = np.linspace(0.1, 1.5, 60)
        prof = [np.asarray(G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=50.0,
                                         abstol=1e-5, reltol=1e-5, ensemble=ens)[1]) for m in (tab_kerr, base)]
    finally:
        ens.set("precision", 64)
    # (measured: 0.6 % of the pixels change class between the two fp32 traces -- as many as between fp64 at 1e-5 and at 1e-9 --,
    # median 4e-7, 99 % 1e-5; against fp64 at 1e-9 1.0 %, 6e-6, 3e-4 for either)
    for other, med, p99, flips in ((imgs["fused"], 5e-6, 1e-3, 0.015), (ref64, 2e-4, 2e-2, 0.02)):
        assert (np.isnan(imgs["tab"]) != np.isnan(other)).sum() <= flips * other.size
        both = ~np.isnan(imgs["tab"]) & ~np.isnan(other)
        assert both.sum() > 5000
        rel = np.abs(imgs["tab"][both] / other[both] - 1)
        assert np.median(rel) < med and np.percentile(rel, 99) < p99, (np.median(rel), np.percentile(rel, 99))
    # (two fp32 traces of a 256 x 256 polar plane at 1e-5: the profiles differ by the step-count noise of their rays, as the sweep
    # points of BASELINE config 5 do -- tests/test_gpu_baseline_configs.py)
    l1 = float(np.abs(prof[0] - prof[1]).sum() / np.abs(prof[1]).sum())
    assert l1 < 5e-2, l1
    print("fp32 line profile through the table against the fused kernels: relative L1", l1)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["kerr-dark-matter", "dilaton-axion"])
def test_segments_and_axis_terms_through_the_fp32_kernels(G, ens, which):
    """The other paths of the table in the fp32 kernels: several radial segments (scalar loads of the segment records) and the
    axis terms of an axion-charged metric, against the same metric's own fp32 kernels."""
    import warnings

    base = {"kerr-dark-matter": G.KerrDarkMatter(1.0, 0.6, 2.0, 8.0, 7.0),
            "dilaton-axion": G.DilatonAxion(1.0, 0.35, 0.16, 0.33)}[which]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tm = G.TabulatedMetric(base)
    x = np.array([0.0, 1000.0, math.radians(20.0 if which == "dilaton-axion" else 70.0), 0.0])
    d = G.ThinDisc(6.0, 60.0)
    chart = G.chart_for_metric(tm, 2000.0)
    kw = dict(pf=G.ConstPointFunctions.affine_time() @ G.ConstPointFunctions.filter_intersected(), image_width=128, image_height=128,
              alpha_lims=(-70, 70), beta_lims=(-70, 70), chart=chart, ensemble=ens)
    ref = G.rendergeodesics(base, x, d, 2000.0, **kw)[2]                      # fp64 at 1e-9
    ens.set("precision", 32)
    try:
        tab32, fused32 = (G.rendergeodesics(m, x, d, 2000.0, abstol=1e-5, reltol=1e-5, **kw)[2] for m in (tm, base))
    finally:
        ens.set("precision", 64)
    # A tolerance of 1e-5 is coarse for these scenes whatever evaluates the metric (fp64 at 1e-5 against fp64 at 1e-9: 2 % of the
    # pixels change class behind the mass shell, 31 % seen from 20 degrees off the axis -- steps longer than the disc's slab is thick,
    # DESIGN.md §5b): the table through the fp32 kernels is held to what the metric's OWN fp32 kernels do against fp64 ...
    def against(img, other):
        both = ~np.isnan(img) & ~np.isnan(other)
        return int((np.isnan(img) != np.isnan(other)).sum()), float(np.median(np.abs(img[both] / other[both] - 1))), int(both.sum())

    flips_t, med_t, n_t = against(tab32, ref)
    flips_f, med_f, n_f = against(fused32, ref)
    assert n_t > 1500 and flips_t <= 1.2 * flips_f + 20 and med_t <= 1.5 * med_f, (flips_t, flips_f, med_t, med_f)
    # ... and, where both fp32 traces hit, to rounding
    _, med, n = against(tab32, fused32)
    assert n > 1500 and med < 2e-6, (med, n)


@pytest.mark.gpu
def test_tabulated_metric_through_the_persistent_kernel_and_ray_arrays(G, ens, tab_kerr):
    """The other launch shapes: a BinningMethod line profile (persistent kernel with wave-ballot refill, four waves per
    workgroup -- four patch caches side by side in LDS) and tracegeodesics on ray arrays in caller order, through the table
    against the fused Kerr kernels."""
    base = tab_kerr.source
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    plane = G.PolarPlane(G.GeometricGrid(), Nr=256, Nθ=256, r_min=1.0, r_max=60.0)
    bins = np.linspace(0.1, 1.5, 60)
    prof = []
    # (the library's own choice for a table is the one-ray-per-lane kernel; the persistent one is a knob away and must agree)
    for m, kern in ((tab_kerr, 1), (tab_kerr, 2), (base, 2)):
        ens.set("kernel", kern)
        xs, ys = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, G.ThinDisc(base.isco(), 50.0), G.BinningMethod(), plane=plane,
                               maxrₑ=50.0, ensemble=ens)
        prof.append(np.asarray(ys))
    ens.set("kernel", 2)
    assert np.max(np.abs(prof[0] - prof[2])) < 1e-7 * np.max(prof[2])
    assert np.max(np.abs(prof[1] - prof[2])) < 1e-7 * np.max(prof[2])
    # ray arrays: 3000 rays in caller order
    rng = np.random.default_rng(3)
    α, β = rng.uniform(-40, 40, 3000), rng.uniform(-25, 25, 3000)
    x = X_FAR
    out = []
    for m in (tab_kerr, base):
        vs = G.map_impact_parameters(m, x, α, β)
        pts = G.tracegeodesics(m, x, vs, G.ThinDisc(3.0, 400.0), 2000.0, ensemble=ens)
        out.append(pts)
    _compare_endpoints(out[0], out[1], x_rtol=1e-7, max_flips=2, max_outliers=3, r_horizon=base.inner_radius())


@pytest.mark.gpu
def test_transfer_functions_of_a_tabulated_metric(G, ens, tab_kerr):
    """Cunningham transfer functions of a user-defined metric: the tangent build of the kernels covers the catalogue, a tabulated
    metric takes the difference-quotient route of the host solvers (safeguarded Newton on ray summaries, central-difference
    Jacobians) -- the same route on the fused Kerr kernel is the comparison."""
    ens.set("kernel", 2).set("precision", 64)
    kerr = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(40), 0.0])
    d = G.ThinDisc(0.0, float("inf"))
    radii = [4.0, 12.0, 40.0]
    a = G.cunningham_transfer_functions(kerr, x, d, radii, N=60, ensemble=ens, root_finder="polished")
    b = G.cunningham_transfer_functions(tab_kerr, x, d, radii, N=60, ensemble=ens, root_finder="polished")
    for ca, cb in zip(a, b):
        ok = np.isfinite(ca.f)
        assert cb.f.size == ca.f.size and ok.sum() >= ca.f.size - 2
        np.testing.assert_array_equal(np.isfinite(cb.f), ok)
        assert cb.gmin == pytest.approx(ca.gmin, rel=1e-7) and cb.gmax == pytest.approx(ca.gmax, rel=1e-7)
        np.testing.assert_allclose(cb.g_star, ca.g_star, atol=1e-9)
        # (difference-quotient Jacobians amplify the table's 1e-9 near the extrema of g, where |∂(ρ, g)/∂(α, β)| -> 0)
        rel = np.abs(cb.f[ok] - ca.f[ok]) / np.max(np.abs(ca.f[ok]))
        assert np.median(rel) < 2e-5 and rel.max() < 2e-2
        sa, sb = float(np.sum((ca.f * ca.g_star)[ok]) / ca.f.size), float(np.sum((cb.f * cb.g_star)[ok]) / cb.f.size)
        assert sb == pytest.approx(sa, rel=1e-4)          # (difference quotients of two traces resolve the jumps at patch edges: the reference route below does not take them)
    # and the tracer says so itself
    from gradus_jl_amd.transfer_functions import device_tracer

    chart = G.chart_for_metric(tab_kerr, 2 * x[1])
    pf = G.ConstPointFunctions.redshift(tab_kerr, x, ensemble=ens)
    assert device_tracer(tab_kerr, x, 2 * x[1], chart, pf, ens).tangent is not None
    # ... and since ABI 8 the reference's own route (Duals through the tracer) is open to a tabulated metric: the same statistics
    c = G.cunningham_transfer_functions(tab_kerr, x, d, radii, N=60, ensemble=ens, root_finder="reference")
    a2 = G.cunningham_transfer_functions(kerr, x, d, radii, N=60, ensemble=ens, root_finder="reference")
    for ca, cc in zip(a2, c):
        ok = np.isfinite(ca.f) & np.isfinite(cc.f)
        assert ok.sum() >= ca.f.size - 2
        assert cc.gmin == pytest.approx(ca.gmin, rel=1e-7) and cc.gmax == pytest.approx(ca.gmax, rel=1e-7)
        sa, sc = float(np.sum((ca.f * ca.g_star)[ok]) / ca.f.size), float(np.sum((cc.f * cc.g_star)[ok]) / cc.f.size)
        # (the Jacobian rides on the table's SECOND derivatives -- cubics on a patch: 1e-5 of scale where |∂(ρ, g)/∂(α, β)| -> 0; the
        # reference's own bound on this statistic is 1e-3 absolute, test/smoke-tests/cunningham-transfer-functions.jl:25-39)
        assert sc == pytest.approx(sa, abs=3e-4)          # (the bar of tests/test_gpu_tangent.py for two builds of one integrator)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 8, 63, 65, 5000])
def test_last_wave_of_a_ray_set_with_few_lanes(G, ens, tab_kerr, n):
    """The wave's cache head in LDS is written by every active lane: a launch whose last wave has fewer lanes than the head has
    entries (5000 rays end in a wave of 8) found stale tags there -- rays that never left the observer.  Traced after other
    kernels have used the CU's LDS, against the fused kernel ray by ray."""
    from gradus_jl_amd.transfer_functions import device_tracer

    ens.set("kernel", 2).set("precision", 64)
    kerr = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(40), 0.0])
    rng = np.random.default_rng(n)
    al, be = rng.uniform(-12, 12, n), rng.uniform(-12, 12, n)
    got = {}
    for name, m in (("kerr", kerr), ("tab", tab_kerr), ("kerr2", kerr), ("tab2", tab_kerr)):
        tr = device_tracer(m, x, 2 * x[1], G.chart_for_metric(m, 2 * x[1]), G.ConstPointFunctions.redshift(m, x, ensemble=ens), ens)
        pts, g = tr(al, be)
        got[name] = (pts["status"].copy(), pts["x"][:, 1].copy(), g)
    for a, b in (("tab", "kerr"), ("tab2", "kerr")):
        np.testing.assert_array_equal(got[a][0], got[b][0])
        hit = got[b][0] == G.StatusCodes.IntersectedWithGeometry
        np.testing.assert_allclose(got[a][1][hit], got[b][1][hit], rtol=1e-7)
        out = hit & (got[b][1] > kerr.isco())
        np.testing.assert_allclose(got[a][2][out], got[b][2][out], rtol=1e-6)
        # (inside the ISCO g rests on the traced plunge and, towards the horizon, on a cancellation in p·u: 2e-4 in 4 of 4879 rays)
        np.testing.assert_allclose(got[a][2][hit & ~out], got[b][2][hit & ~out], rtol=1e-3)
    np.testing.assert_array_equal(got["tab"][1], got["tab2"][1])


def test_a_metric_the_table_cannot_represent_is_refused_loudly(G):
    """A discontinuity inside the radial range (Kerr with a step in M at r = 9): the fit's own estimates do not come down, the
    constructor says so instead of handing the kernels a table of garbage; strict=False warns and goes on."""
    kerr_in, kerr_out = G.KerrMetric(1.0, 0.5), G.KerrMetric(1.02, 0.5)

    def stepped(r, th):
        a, b = kerr_in._components(r, np.sin(th), np.cos(th)), kerr_out._components(r, np.sin(th), np.cos(th))
        return tuple(np.where(np.asarray(r) < 9.0, x, y) for x, y in zip(a, b))

    with pytest.raises(ValueError, match="not smooth"):
        G.TabulatedMetric(stepped, inner_radius=kerr_in.inner_radius(), isco=kerr_in.isco(), r_max=200.0)
    with pytest.warns(UserWarning, match="not smooth"):
        tm = G.TabulatedMetric(stepped, inner_radius=kerr_in.inner_radius(), isco=kerr_in.isco(), r_max=200.0, strict=False)
    assert tm.errors[0] > 1e-6 and (tm.m_r, tm.n_theta) == (36, 144)      # one refinement showed no convergence: no further ones
    # ... and the same function with its break NAMED is an ordinary table on the default grid: no patch straddles r = 9
    tb = G.TabulatedMetric(stepped, inner_radius=kerr_in.inner_radius(), isco=kerr_in.isco(), r_max=200.0, breaks=[9.0])
    assert (tb.m_r, tb.n_theta) == (24, 96) and tb.errors[0] < 1e-10 and tb.errors[1] < 1e-7 and tb.grid.n_seg == 2
    for r in (8.9999999, 9.0000001, 3.0, 150.0):
        g, dr, _ = tb.table_jacobian(r, 1.1)
        ref = (kerr_in if r < 9.0 else kerr_out).metric_components(r, 1.1)
        np.testing.assert_allclose(g, ref, rtol=2e-9)


def test_an_inner_radius_inside_the_horizon_moves_the_table_out(G):
    """`inner_radius` below the outermost horizon (the reference's dilaton-axion and Kerr-dark-matter formulas do that): the
    pole of g_rr would lie inside the table; the range -- and the chart's inner boundary -- start at the sign change instead."""
    kerr = G.KerrMetric(1.0, 0.6)
    f = lambda r, th: kerr._components(r, np.sin(th), np.cos(th))
    with pytest.warns(UserWarning, match="g_rr changes sign"):
        tm = G.TabulatedMetric(f, inner_radius=0.8 * kerr.inner_radius(), isco=kerr.isco(), r_max=500.0)
    assert tm.inner_radius() == pytest.approx(kerr.inner_radius(), rel=1e-9)
    assert tm.errors[0] < 1e-10 and tm.errors[1] < 1e-7


def test_axion_charge_takes_the_axis_form(G):
    """g_ϕϕ and g_tϕ of a dilaton-axion metric with β != 0 do NOT vanish on the polar axis (W ∝ csc²θ, dilaton-axion-ad.jl:13-14):
    divided by sin²θ they are singular, as sampled they pass through zero a few milliradians off the axis and a polynomial's absolute
    error is no relative one there.  The constructor finds that out from the fit and stores them as K_m(r) + K_d(r) cos θ +
    sin²θ h(r, θ) (gr_metric_grid.pole_factor = 2): the limits on the two poles as polynomials of their own, the rest smooth.
    The table then follows the metric to 1e-9 of ITSELF next to the axis, on both poles; with β = 0 nothing changes."""
    import warnings

    da = G.DilatonAxion(1.0, 0.35, 0.16, 0.33)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (the inner radius moves out to the horizon: tested above)
        tm = G.TabulatedMetric(da)
        t0 = G.TabulatedMetric(G.DilatonAxion(1.0, 0.5, 0.0, 1.0))
    assert tm.grid.pole_factor == 2 and t0.grid.pole_factor == 1
    assert (tm.m_r, tm.n_theta) == (24, 96) and tm.errors[0] < 1e-10 and tm.errors[1] < 1e-7 and tm.errors[2] < 1e-7
    for r in (2.9, 4.0, 40.0, 900.0):
        for th in (1e-4, 3e-3, 0.03, 0.05, 1.3, math.pi - 0.02, math.pi - 2e-3):
            g, dr, dth = tm.table_jacobian(r, th)
            ref = np.array(da.metric_components(r, th))
            np.testing.assert_allclose(g, ref, rtol=1e-9)
            # the derivatives of the azimuthal components against difference quotients of the metric
            h = 1e-6
            fd_r = (np.array(da.metric_components(r * (1 + h), th)) - np.array(da.metric_components(r * (1 - h), th))) / (2 * h * r)
            np.testing.assert_allclose(dr[3:], fd_r[3:], rtol=3e-6, atol=1e-9)
    # the two poles have different limits (the metric is not symmetric about the equator): K_d is not zero
    gN, gS = tm.table_jacobian(4.0, 1e-5)[0], tm.table_jacobian(4.0, math.pi - 1e-5)[0]
    assert abs(gN[3] - gS[3]) > 0.1 and abs(gN[4] - gS[4]) > 0.1
    # a Jet through the table's own polynomials (the generic ISCO's route) includes the axis terms
    np.testing.assert_allclose(tm._table_components(4.0, 0.01)[3], da.metric_components(4.0, 0.01)[3], rtol=1e-9)


def _as_callable(m):
    """a catalogue metric stripped to the plugin contract: metric_components and nothing else"""
    return lambda r, th: m._components(r, np.sin(th), np.cos(th))


def test_piecewise_metrics_pass_the_fit_once_their_breaks_are_named(G):
    """KerrDarkMatter (kerr-dark-matter.jl:12-20: the enclosed mass is piecewise, kinks at rₛ and rₛ + Δr) and KerrRefractive
    (kerr-refractive-ad.jl:26 + utils.jl:158-168: jumps of 6e-5 in n at corona_radius ± 1.25, an arctangent step 2.5e-4 wide at
    corona_radius) as bare CALLABLES: refused without their break radii, ordinary tables on the default grid with them -- a segment
    starts at every break, patches shrink geometrically towards the step from both sides."""
    kdm = G.KerrDarkMatter(1.0, 0.6, 2.0, 8.0, 7.0)      # (a mass shell narrow enough that the fit sees the kinks)
    with pytest.raises(ValueError, match="not smooth"):
        G.TabulatedMetric(_as_callable(kdm), inner_radius=kdm.inner_radius(), isco=6.0)
    tm = G.TabulatedMetric(_as_callable(kdm), inner_radius=kdm.inner_radius(), isco=6.0, breaks=[7.0, 15.0])
    assert (tm.m_r, tm.n_theta) == (24, 96) and tm.grid.n_seg == 3 and tm.errors[0] < 1e-10 and tm.errors[1] < 1e-7
    for r in (3.0, 6.999999, 7.000001, 11.0, 14.999999, 15.000001, 400.0):
        g, dr, _ = tm.table_jacobian(r, 0.9)
        np.testing.assert_allclose(g, kdm.metric_components(r, 0.9), rtol=2e-9)
        h = 1e-7 * r
        fd = (np.array(kdm.metric_components(r + h, 0.9)) - np.array(kdm.metric_components(r - h, 0.9))) / (2 * h)
        np.testing.assert_allclose(dr, fd, rtol=2e-5, atol=1e-8)
    # the catalogue type names its own breaks
    assert G.TabulatedMetric(kdm).grid.n_seg == 3
    kr = G.KerrRefractive(1.0, 0.5, 1.2, 20.0)
    with pytest.raises(ValueError, match="not smooth"):
        G.TabulatedMetric(_as_callable(kr), inner_radius=kr.inner_radius(), isco=kr.isco())
    tr = G.TabulatedMetric(_as_callable(kr), inner_radius=kr.inner_radius(), isco=kr.isco(), breaks=kr.break_radii())
    assert (tr.m_r, tr.n_theta) == (24, 96) and tr.grid.n_seg == 5 and tr.errors[0] < 1e-10 and tr.errors[1] < 1e-7
    for r in (5.0, 18.7499, 18.7501, 19.9, 19.999, 19.99999, 20.0, 20.00001, 20.001, 20.4, 21.2499, 21.2501, 300.0):
        g, dr, _ = tr.table_jacobian(r, 1.2)
        np.testing.assert_allclose(g, kr.metric_components(r, 1.2), rtol=2e-9)
    # across the arctangent step g_tt changes by a factor n² = 1.44 within 1e-3: the table resolves its slope
    g1, d1, _ = tr.table_jacobian(20.0, 1.2)
    fd = (kr.metric_components(20.0 + 1e-8, 1.2)[0] - kr.metric_components(20.0 - 1e-8, 1.2)[0]) / 2e-8
    assert abs(d1[0] / fd - 1.0) < 1e-5 and abs(fd) > 100.0


def test_breaks_are_validated_and_too_many_are_refused(G):
    L = G._lib
    lib = L.load()
    grid = L.gr_metric_grid()
    mk = lambda *bs: (L.gr_metric_break * len(bs))(*[L.gr_metric_break(r, s) for r, s in bs])
    assert lib.gr_metric_grid_plan_breaks(2.0, 100.0, 1.9, 8, 32, 1, mk((1.5, 0.0)), grid) == -1          # outside (r_min, r_max)
    assert lib.gr_metric_grid_plan_breaks(2.0, 100.0, 1.9, 8, 32, 2, mk((5.0, 0.0), (5.0, 0.0)), grid) == -1   # twice the same
    assert lib.gr_metric_grid_plan_breaks(2.0, 100.0, 1.9, 8, 32, 1, mk((5.0, -1.0)), grid) == -1
    assert lib.gr_metric_grid_plan_breaks(2.0, 100.0, 1.9, 8, 32, 1, None, grid) == -1
    many = mk(*[(3.0 + k, 0.0) for k in range(L.GR_METRIC_MAX_SEG)])
    assert lib.gr_metric_grid_plan_breaks(2.0, 100.0, 1.9, 8, 32, len(many), many, grid) == -1 and b"GR_METRIC_MAX_SEG" in lib.gr_last_error()
    # any order; every break starts a segment anchored there, behind a core of m_r parts
    assert lib.gr_metric_grid_plan_breaks(2.0, 100.0, 1.9, 8, 32, 2, mk((30.0, 0.0), (10.0, 0.0)), grid) == 0
    assert grid.n_seg == 3 and [grid.seg[k].r_lo for k in range(3)] == [2.0, 10.0, 30.0]
    assert grid.seg[1].anchor == 10.0 and grid.seg[1].core == 1 and grid.seg[0].core == 0 and grid.seg[0].anchor == 1.9
    assert sum(grid.seg[k].n_rows for k in range(3)) == grid.n_rows and grid.n_r_nodes == grid.n_rows * grid.fit_nodes
    rn, tn = np.empty(grid.n_r_nodes), np.empty(grid.n_theta_nodes)
    assert lib.gr_metric_grid_nodes(grid, rn.ctypes.data, tn.ctypes.data) == 0
    # no node of a segment lies beyond the break that ends it
    for k in range(3):
        sg = grid.seg[k]
        nodes = rn[sg.first_row * grid.fit_nodes:(sg.first_row + sg.n_rows) * grid.fit_nodes]
        assert nodes.min() >= sg.fit_lo and nodes.max() <= sg.fit_hi


def test_negative_radii_through_a_wormhole_throat(G):
    """A chart that continues through the throat of the Morris-Thorne wormhole (morris-thorne-ad.jl: l runs over both signs):
    r_min < 0 and a break at 0 of the throat's size -- segments measure distances from their anchors, not radii."""
    mt = G.MorrisThorneWormhole(1.0)
    tm = G.TabulatedMetric(mt, r_min=-1000.0, r_max=1000.0, breaks=[(0.0, 1.0)])
    assert tm.grid.pole_factor == 0 and tm.grid.n_seg == 3 and tm.errors[0] < 1e-10 and tm.errors[1] < 1e-7
    for r in (-900.0, -256.5, -100.0, -3.0, -1e-3, 0.0, 1e-5, 0.3, 2.0, 77.0, 999.0):
        g, dr, dth = tm.table_jacobian(r, 0.7)
        np.testing.assert_allclose(g, mt.metric_components(r, 0.7), rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(dr[2], 2.0 * r, rtol=1e-10, atol=1e-10)          # ∂r g_θθ = 2 l: changes sign in the throat


def test_the_table_grows_to_cover_a_chart_and_nothing_is_extrapolated(G):
    """polynomials do not extrapolate -- g_tt = -3e4 at ten times r_max.  `cover` refits on a range that
    contains the chart; the radial clamps keep NaN and ±inf (a trial step that overflowed) inside the table."""
    tm = G.TabulatedMetric(G.KerrMetric(1.0, 0.5), r_max=100.0)
    assert tm.r_max == 100.0
    tm.cover(tm.r_min, 100.0)
    assert tm.r_max == 100.0                       # nothing to do
    id0 = tm.table[8]
    tm.cover(tm.r_min, 6000.0)
    assert tm.r_max == 12000.0 and tm.table[8] != id0 and tm.table[13] == 12000.0      # H_BUILD_ID, H_RMAX
    np.testing.assert_allclose(tm.table_jacobian(5000.0, 1.0)[0], tm.source.metric_components(5000.0, 1.0), rtol=2e-9)
    with pytest.raises(ValueError, match="crosses the horizon"):
        tm.cover(0.5 * tm.inner_radius(), 100.0)
    for r in (float("inf"), float("-inf"), float("nan"), 1e300):
        g, dr, dth = tm.table_jacobian(r, 1.0)
        assert np.all(np.isfinite(g)) and np.all(np.isfinite(dr))


@pytest.mark.gpu
def test_dilaton_axion_through_the_table_equals_its_fused_kernel(G, ens):
    """A catalogue metric with an axion charge through the table (raw azimuthal components, range starting at the horizon) against
    its hand-fused kernel: every pixel of a small image whose rays end away from the hole."""
    import warnings

    ens.set("kernel", 2).set("precision", 64)
    base = G.DilatonAxion(1.0, 0.35, 0.16, 0.33)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tm = G.TabulatedMetric(base)
    x = np.array([0.0, 500.0, math.radians(60), 0.0])
    d = G.ThinDisc(0.0, 40.0)
    kw = dict(image_width=96, image_height=64, alpha_lims=(-25, 25), beta_lims=(-15, 15), ensemble=ens)
    ref = G.prerendergeodesics(base, x, d, 1000.0, **kw)[2].points.ravel()
    got = G.prerendergeodesics(tm, x, d, 1000.0, **kw)[2].points.ravel()
    lost = lambda st: (st == G.StatusCodes.WithinInnerBoundary) | (st == G.StatusCodes.NoStatus)
    same = (ref["status"] == got["status"]) | (lost(ref["status"]) & lost(got["status"])) | (lost(got["status"]) & (ref["x"][:, 1] < 1.03 * tm.inner_radius()))
    assert (~same).sum() <= 2
    cmp_ = (ref["status"] == got["status"]) & ~lost(ref["status"])
    assert cmp_.sum() > 4000
    np.testing.assert_allclose(got["x"][cmp_][:, 1:3], ref["x"][cmp_][:, 1:3], rtol=1e-6, atol=1e-6)


def test_isco_of_a_bare_callable_comes_from_the_table(G):
    """A callable brings no `isco`: the generic ISCO search (special-radii.jl:14-60: a downward scan of E(r) over 19 800 radii,
    then a bracketing root find of dE/dr) runs on the table's own polynomials -- every radius of the scan in its own patch."""
    kerr = G.KerrMetric(1.0, 0.9)
    tm = G.TabulatedMetric(lambda r, th: kerr._components(r, np.sin(th), np.cos(th)), inner_radius=kerr.inner_radius())
    assert tm.isco() == pytest.approx(kerr.isco(), rel=5e-7)          # (second derivatives of the table's polynomials)
    rs = np.array([1.6, 2.5, 7.0, 90.0])
    got = tm._table_components(rs, 1.2)
    for k in range(5):
        np.testing.assert_allclose(got[k], [kerr.metric_components(r, 1.2)[k] for r in rs], rtol=1e-9)
    # what the corona's disc kinematics evaluate on arrays of bin radii
    np.testing.assert_allclose(G.corona.circular_fourvelocity(tm, rs[1:]), G.corona.circular_fourvelocity(kerr, rs[1:]), rtol=1e-7)
    np.testing.assert_allclose(G.corona._proper_area(tm, rs, math.pi / 2), G.corona._proper_area(kerr, rs, math.pi / 2), rtol=1e-9)


@pytest.mark.gpu
def test_line_profile_of_a_tabulated_metric(G, ens, tab_kerr):
    """lineprofile(bins, ε, m, u, d, BinningMethod()) fused on the device (C5's route: separable polar plane, LDS histogram,
    persistent kernel) through the table against the fused Kerr kernel, for a power law and for an emissivity profile that itself
    came from a corona traced through the table."""
    ens.set("kernel", 2).set("precision", 64)
    kerr = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(0.0, 400.0)
    bins = np.linspace(0.1, 1.5, 120)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=256, Nθ=256, r_max=60.0)
    kw = dict(plane=plane, maxrₑ=50.0, callback=G.domain_upper_hemisphere(), ensemble=ens)
    _, fa = G.lineprofile(bins, G.PowerLawEmissivity(3), kerr, x, d, G.BinningMethod(), **kw)
    _, fb = G.lineprofile(bins, G.PowerLawEmissivity(3), tab_kerr, x, d, G.BinningMethod(), **kw)
    assert fa.sum() == pytest.approx(1.0) and np.count_nonzero(fa) > 60
    np.testing.assert_allclose(fb, fa, atol=2e-6 * fa.max())
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    pa = G.emissivity_profile(kerr, d, G.LampPostModel(h=6.0), sampler=s, n_samples=20_000, N=60, ensemble=ens)
    pb = G.emissivity_profile(tab_kerr, d, G.LampPostModel(h=6.0), sampler=s, n_samples=20_000, N=60, ensemble=ens)
    _, ga = G.lineprofile(bins, pa, kerr, x, d, G.BinningMethod(), **kw)
    _, gb = G.lineprofile(bins, pb, tab_kerr, x, d, G.BinningMethod(), **kw)
    np.testing.assert_allclose(gb, ga, atol=1e-4 * ga.max())


def test_charged_test_particles_are_not_traced_through_a_table(G):
    kn = G.KerrNewmanMetric(1.0, 0.5, 0.3)
    tm = G.TabulatedMetric(kn, m_r=4, n_theta=8, max_refinements=0, strict=False, r_max=200.0)
    x = np.array([0.0, 100.0, 1.2, 0.0])
    with pytest.raises(NotImplementedError, match="charged test particles"):
        G.tracing_configuration(tm, x, np.zeros((1, 4)), None, 200.0, q=0.5, μ=1.0, ensemble=object.__new__(G.EnsembleMI355X))
